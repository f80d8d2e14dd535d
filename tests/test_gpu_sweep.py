"""GPU sweep over dimensions at the seams between layouts (lane / coop / wave / split) and over all five
integrators: two iterations of three chains each, draws and step counts against the oracle."""
import numpy as np
import pytest

import oracle_lib as ol
from test_gpu_parity import SEED, close, dense_problem, wa  # noqa: F401

pytestmark = pytest.mark.gpu

DIMS_DIAG = [1, 4, 5, 24, 25, 64, 65, 128, 129, 256, 257, 512, 513, 1024, 1025, 2048, 4096]
DIMS_DENSE = [2, 16, 17, 64, 100, 101, 128, 129, 500]
INTEG = [ol.I_FIXED, ol.I_ADAPT_D, ol.I_ADAPT_R2P, ol.I_ADAPT_YOSHIDA, ol.I_ADAPT_FLOWD]


def _run(wa, tk, D, ig, tpar, q0, M=4, numIter=2):
    e = wa.Engine(D, q0.shape[0], integrator=ig, H0=0.7, delta0=0.2, M=M, chain_offset=11)
    e.set_target(tk, tpar)
    s, d = e.run(q0, SEED, numIter)
    s = s.transpose(1, 2).cpu().numpy(); d = d.cpu().numpy(); name = e.kernel_name; e.close()
    for k in (0, q0.shape[0] - 1):
        o = ol.walnuts(tk, q0[k], integrator=ig, tpar=tpar, H0=0.7, delta0=0.2, numIter=numIter, M=M, seed=SEED, chain=11 + k)
        np.testing.assert_array_equal(d[k][:, [6, 7, 19, 20]], o["diagnostics"][:, [6, 7, 19, 20]], err_msg=f"{name} D={D} ig={ig}")
        close(s[k], o["samples"], what=f"{name} D={D} ig={ig} chain {k}")


@pytest.mark.parametrize("D", DIMS_DIAG)
def test_diag_gaussian_every_layout_seam(wa, D):
    rng = np.random.default_rng(D)
    a = 1.0 / np.logspace(0, 1.5, D)
    q0 = rng.normal(size=(3, D)) / np.sqrt(a)
    for ig in INTEG:
        _run(wa, ol.T_DIAGGAUSS, D, ig, a, q0)


@pytest.mark.parametrize("D", DIMS_DENSE)
def test_dense_gaussian_every_layout_seam(wa, D):
    A, q0 = dense_problem(D, 30.0, 3, 100 + D)
    for ig in INTEG:
        _run(wa, ol.T_DENSEGAUSS, D, ig, A.ravel(), q0)


def test_unsupported_sizes_fail_loudly(wa):
    with pytest.raises(NotImplementedError):
        wa.Engine(262145, 1).set_target(ol.T_STDGAUSS)
    with pytest.raises(NotImplementedError):  # the cluster of a chain must be resident: n_chains * ceil(D/4096) <= CUs
        wa.Engine(100000, 64).set_target(ol.T_STDGAUSS)
    with pytest.raises(NotImplementedError):
        wa.Engine(1030, 2).set_target(ol.T_DENSEGAUSS, np.eye(1030).ravel())
    with pytest.raises(ValueError):
        wa.Engine(12, 2).set_target(ol.T_FUNNEL10)


@pytest.mark.parametrize("C", [1, 63, 64, 65, 129])
def test_split_ragged_chain_counts(wa, C):
    """The gradient kernel works on 64-chain tiles; chain counts around the tile edges."""
    D = 140
    A, q0 = dense_problem(D, 30.0, C, 7)
    e = wa.Engine(D, C, integrator=ol.I_FIXED, H0=0.7, M=5, chain_offset=3).set_target(ol.T_DENSEGAUSS, A.ravel())
    s, d = e.run(q0, SEED, 2)
    s = s.transpose(1, 2).cpu().numpy(); d = d.cpu().numpy(); e.close()
    for k in sorted({0, C // 2, C - 1}):
        o = ol.walnuts(ol.T_DENSEGAUSS, q0[k], integrator=ol.I_FIXED, tpar=A.ravel(), H0=0.7, numIter=2, M=5, seed=SEED, chain=3 + k)
        np.testing.assert_array_equal(d[k][:, [6, 7, 19, 20]], o["diagnostics"][:, [6, 7, 19, 20]])
        close(s[k], o["samples"], what=f"C={C} chain {k}")


@pytest.mark.parametrize("layout", ["wave", "split"])
def test_deep_trees_in_wave_and_split(wa, layout):
    """M = 10 with a small step size: orbits of up to 1023 steps exercise every slot level of the memory-resident
    sub-tree store (the one-chain-per-wave layouts cache none on chip)."""
    D = 300 if layout == "wave" else 150
    rng = np.random.default_rng(2)
    if layout == "wave":
        tk, tpar = ol.T_DIAGGAUSS, np.ones(D); q0 = rng.normal(size=(2, D))
    else:
        tk = ol.T_DENSEGAUSS; A, q0 = dense_problem(D, 4.0, 2, 9); tpar = A.ravel()
    e = wa.Engine(D, 2, integrator=ol.I_FIXED, H0=0.01, M=10, chain_offset=5).set_target(tk, tpar)
    assert layout in e.kernel_name
    s, d = e.run(q0, SEED, 2)
    s = s.transpose(1, 2).cpu().numpy(); d = d.cpu().numpy(); e.close()
    assert d[:, :, 20].max() >= 9  # deep doublings were reached
    for k in (0, 1):
        o = ol.walnuts(tk, q0[k], integrator=ol.I_FIXED, tpar=tpar, H0=0.01, numIter=2, M=10, seed=SEED, chain=5 + k)
        np.testing.assert_array_equal(d[k][:, [6, 7, 19, 20]], o["diagnostics"][:, [6, 7, 19, 20]])
        close(s[k], o["samples"], what=f"{layout} chain {k}")


def test_headline_config_full_size_spot_check(wa):
    """BASELINE config 2 at its full size (4096 chains, 100-D kappa = 1e4 dense Gaussian, fixedLeapFrog, M = 10, H0 = 0.5):
    32 chains spread over the launch, three iterations each, against the oracle chain by chain."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import make_dense_problem, SEED as BSEED
    C, D = 4096, 100
    A, q0 = make_dense_problem(D, 1e4, C, BSEED)
    e = wa.Engine(D, C, integrator=ol.I_FIXED, H0=0.5, M=10).set_target(ol.T_DENSEGAUSS, A)
    s, d = e.run(q0, BSEED, 3)
    s = s.transpose(1, 2).cpu().numpy(); d = d.cpu().numpy(); e.close()
    rng = np.random.default_rng(0)
    for k in sorted(set(rng.integers(0, C, size=30).tolist() + [0, C - 1])):
        o = ol.walnuts(ol.T_DENSEGAUSS, q0[k], integrator=ol.I_FIXED, tpar=A.ravel(), H0=0.5, numIter=3, M=10, seed=BSEED, chain=k)
        np.testing.assert_array_equal(d[k][:, [6, 7, 19, 20]], o["diagnostics"][:, [6, 7, 19, 20]], err_msg=f"chain {k}")
        close(s[k], o["samples"], what=f"chain {k}")
        close(d[k], o["diagnostics"], what=f"diagnostics chain {k}")


@pytest.mark.parametrize("D,ig", [(5000, ol.I_FIXED), (9000, ol.I_ADAPT_R2P), (20000, ol.I_ADAPT_D), (70000, ol.I_FIXED)])
def test_cluster_layout_huge_gaussians(wa, D, ig):
    """D > 4096 (the reference's Gaussian scaling scripts go to 2^18): a cluster of ceil(D/4096) workgroups per
    chain, vectors in registers, reductions across workgroups through global memory."""
    rng = np.random.default_rng(D)
    q0 = rng.normal(size=(2, D))
    H0 = 1.4 * D ** -0.25
    e = wa.Engine(D, 2, integrator=ig, H0=H0, delta0=0.3, M=5, generated=ol.G_Q0_SUMSQ, chain_offset=21).set_target(ol.T_STDGAUSS)
    assert "cluster" in e.kernel_name
    s, d = e.run(q0, SEED, 3)
    s = s.transpose(1, 2).cpu().numpy(); d = d.cpu().numpy(); e.close()
    for k in (0, 1):
        o = ol.walnuts(ol.T_STDGAUSS, q0[k], integrator=ig, H0=H0, delta0=0.3, numIter=3, M=5, seed=SEED, chain=21 + k,
                       generated=ol.G_Q0_SUMSQ)
        np.testing.assert_array_equal(d[k][:, [6, 7, 19, 20]], o["diagnostics"][:, [6, 7, 19, 20]])
        close(s[k], o["samples"], what=f"D={D} chain {k}")


def test_cluster_abort_path_returns_an_error_without_hanging(wa):
    """A workgroup of a cluster that never arrives at a cross-workgroup reduction (test hook
    wn_debug_cluster_drop = the exchange its last slice skips): the siblings' bounded spins give up, set the
    abort flag, every loop of every workgroup ends and wn_run reports an error."""
    import time
    D = 9000
    q0 = np.random.default_rng(1).normal(size=(2, D))
    e = wa.Engine(D, 2, integrator=ol.I_FIXED, H0=0.1, M=4, generated=ol.G_Q0_SUMSQ).set_target(ol.T_STDGAUSS)
    assert "cluster=3" in e.kernel_name
    assert e.lib.wn_debug_cluster_drop(e.h, 7) == 0
    t0 = time.time()
    with pytest.raises(wa.WalnutsError, match="timed out"):
        e.run(q0, SEED, 3)
    assert time.time() - t0 < 60.0
    assert e.lib.wn_debug_cluster_drop(e.h, 0) == 0
    s, d = e.run(q0, SEED, 3)  # the engine is usable again: counters and the abort flag are reset per launch
    o = ol.walnuts(ol.T_STDGAUSS, q0[1], integrator=ol.I_FIXED, H0=0.1, numIter=3, M=4, seed=SEED, chain=1, generated=ol.G_Q0_SUMSQ)
    close(s.transpose(1, 2).cpu().numpy()[1], o["samples"], what="after an aborted launch")
    e.close()


def test_gaussian_scaling_script_call_shape(wa):
    """The call of WALNUTSpy_examples/gaussian/mainGaussESS.py:74-80 at a dimension of its upper range."""
    from walnuts_amd import adaptiveIntegrators as ai, targetDistr as td
    d = 2 ** 16
    q0 = np.random.default_rng(0).normal(size=d)
    for integ in (ai.fixedLeapFrog, ai.adaptLeapFrogR2P):
        samples, diagnostics = wa.WALNUTS(td.stdGauss, q0, generated=wa.q0_sumsq, integrator=integ, M=10,
                                          H0=1.4 * d ** -0.25, numIter=6, warmupIter=0, seed=5)
        assert samples.shape == (2, 7) and diagnostics.shape == (6, 24)
        assert abs(samples[1, 1:].mean() / d - 1.0) < 0.02  # sum q^2 ~ chi2_d


def test_cluster_chunks_more_chains_than_fit(wa):
    """C * ceil(D/4096) > 256: WALNUTS() runs the chains in consecutive launches; same draws as one chain at a time."""
    from walnuts_amd import adaptiveIntegrators as ai, targetDistr as td
    D, C = 140000, 9  # 35 workgroups per chain: 7 chains per launch
    q0 = np.random.default_rng(3).normal(size=(C, D))
    kw = dict(generated=wa.q0_sumsq, integrator=ai.fixedLeapFrog, M=4, H0=1.4 * D ** -0.25, numIter=3, warmupIter=0, seed=8)
    s, d = wa.WALNUTS(td.stdGauss, q0, **kw)
    assert s.shape == (C, 2, 4) and d.shape == (C, 3, 24)
    s1, d1 = wa.WALNUTS(td.stdGauss, q0[8], chain_offset=8, **kw)
    np.testing.assert_array_equal(s[8], s1); np.testing.assert_array_equal(d[8], d1)


def test_cluster_layout_warmup(wa):
    """Warm-up adaptation in the cluster layout: every workgroup of a chain keeps its own sorted quantile store."""
    D = 9000
    q0 = np.random.default_rng(4).normal(size=(2, D))
    H0 = 1.4 * D ** -0.25
    e = wa.Engine(D, 2, integrator=ol.I_ADAPT_R2P, H0=H0, delta0=0.3, M=5, generated=ol.G_Q0_SUMSQ, chain_offset=2,
                  warmupIter=12).set_target(ol.T_STDGAUSS)
    s, d = e.run(q0, SEED, 16)
    s = s.transpose(1, 2).cpu().numpy(); d = d.cpu().numpy(); e.close()
    for k in (0, 1):
        o = ol.walnuts(ol.T_STDGAUSS, q0[k], integrator=ol.I_ADAPT_R2P, H0=H0, delta0=0.3, numIter=16, M=5, seed=SEED,
                       chain=2 + k, generated=ol.G_Q0_SUMSQ, warmupIter=12)
        close(d[k][:, [15, 18]], o["diagnostics"][:, [15, 18]], what=f"H/delta chain {k}")
        np.testing.assert_array_equal(d[k][:, [6, 7, 19, 20]], o["diagnostics"][:, [6, 7, 19, 20]])
        close(s[k], o["samples"], what=f"chain {k}")


@pytest.mark.parametrize("tk,D", [(ol.T_DIAGGAUSS, 5), (ol.T_DIAGGAUSS, 16), (ol.T_DIAGGAUSS, 24), (ol.T_DENSEGAUSS, 8), (ol.T_DENSEGAUSS, 16)])
def test_lane_layout_on_request_and_by_chain_count(wa, tk, D):
    """Small Gaussians run one chain per lane only when asked (waves_per_group = 64) or when there are enough
    chains to fill the chip (>= 8192 with fixedLeapFrog); both choices give the oracle's draws."""
    rng = np.random.default_rng(D)
    if tk == ol.T_DENSEGAUSS:
        A, q0 = dense_problem(D, 30.0, 70, 50 + D); tpar = A.ravel()
    else:
        tpar = 1.0 / np.logspace(0, 1.0, D); q0 = rng.normal(size=(70, D)) / np.sqrt(tpar)
    for ig in (ol.I_FIXED, ol.I_ADAPT_R2P):
        e = wa.Engine(D, 70, integrator=ig, H0=0.7, delta0=0.2, M=4, chain_offset=11, waves_per_group=64).set_target(tk, tpar)
        assert "lane" in e.kernel_name
        s, d = e.run(q0, SEED, 2)
        s = s.transpose(1, 2).cpu().numpy(); d = d.cpu().numpy(); e.close()
        for k in (0, 69):
            o = ol.walnuts(tk, q0[k], integrator=ig, tpar=tpar, H0=0.7, delta0=0.2, numIter=2, M=4, seed=SEED, chain=11 + k)
            np.testing.assert_array_equal(d[k][:, [6, 7, 19, 20]], o["diagnostics"][:, [6, 7, 19, 20]])
            close(s[k], o["samples"], what=f"lane D={D} ig={ig} chain {k}")
    big = wa.Engine(D, 8192, integrator=ol.I_FIXED, H0=0.7, M=4).set_target(tk, tpar)
    small = wa.Engine(D, 4096, integrator=ol.I_FIXED, H0=0.7, M=4).set_target(tk, tpar)
    assert "lane" in big.kernel_name and ("coop" in small.kernel_name or "wn_fast" in small.kernel_name)
    big.close(); small.close()


def test_randomised_differential_fuzz():
    """tools/fuzz_parity.py: random (target, D, integrator, M, step size, warm-up, generated, chain count) drawn
    over every layout; step counts must equal the oracle's exactly, draws to rtol 1e-6 of their vector's scale."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "120", "11"], capture_output=True,
                       text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "120 configurations, 0 mismatching chains" in r.stdout, r.stdout[-3000:]
