"""BASELINE.json's configurations at the sizes bench.py runs them, through the C ABI: size-independent step-accounting
properties on every chain plus a sample of chains against the oracle (integers exact, floats rtol 1e-6).
cfg 2 (headline, 4096 chains x 12 iterations, 16 chains vs oracle) is tests/test_gpu_fast.py."""
import os

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
SEED = 20251121
INT_COLS = [6, 7, 19, 20]


@pytest.fixture(scope="module")
def wa():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import walnuts_amd
    return walnuts_amd


def close(a, b, rtol=1e-6, atol=1e-9, what=""):
    a = np.asarray(a, float); b = np.asarray(b, float)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    with np.errstate(all="ignore"):
        ok = (np.isnan(a) & np.isnan(b)) | (np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))) | \
             (np.abs(a - b) <= atol + rtol * np.abs(b))
    assert ok.all(), f"{what}: {np.count_nonzero(~ok)} mismatches, first at {np.argwhere(~ok)[:4].tolist()}"


def test_cfg3_funnel_r2p_4096_chains_m12(wa):
    """SURVEY 8d cfg 3: funnel10, adaptLeapFrogR2P, M = 12, H0 = 0.3, delta0 = 0.3, 4096 chains, 20 iterations."""
    C, IT = 4096, 20
    rng = np.random.default_rng([SEED, 3, 0])
    q0 = np.zeros((C, 11)); q0[:, 0] = 3.0 * rng.normal(size=C)
    q0[:, 1:] = np.exp(0.5 * q0[:, :1]) * rng.normal(size=(C, 10))
    e = wa.Engine(11, C, integrator=ol.I_ADAPT_R2P, H0=0.3, delta0=0.3, M=12, generated=ol.G_FIRSTK, gen_k=2).set_target(ol.T_FUNNEL10)
    s, d = e.run(q0, SEED, IT)
    s = s.transpose(1, 2).cpu().numpy(); d = d.cpu().numpy(); e.close()
    assert np.isfinite(s).all()
    code = d[:, :, 19]
    assert np.isin(code, [0, 4, -4, 5, 999]).all()
    ok = np.isin(code, [0, 4, -4])  # (a sub-U-turn, code 5, stops inside the last doubling)
    # every orbit whose last doubling completed: 2^NdoublComputed - 1 macro steps, each at least one forward evaluation; R2P's c in [0, maxC+1]
    macro = 2.0 ** d[:, :, 20] - 1
    assert (d[:, :, 6][ok] >= macro[ok]).all() and (d[:, :, 7] >= 0).all()
    assert (d[:, :, 21] >= 0).all() and (d[:, :, 22] <= 11).all() and (d[:, :, 1] <= d[:, :, 20]).all()
    assert (d[:, :, 4] - d[:, :, 5] == macro)[ok].all()  # maxFint - maxBint = number of macro steps
    for k in sorted(set(np.random.default_rng(1).integers(0, C, size=14).tolist() + [0, C - 1])):
        o = ol.walnuts(ol.T_FUNNEL10, q0[k], integrator=ol.I_ADAPT_R2P, H0=0.3, delta0=0.3, numIter=IT, M=12, seed=SEED, chain=k,
                       generated=ol.G_FIRSTK, gen_k=2)
        np.testing.assert_array_equal(d[k][:, INT_COLS], o["diagnostics"][:, INT_COLS], err_msg=f"chain {k}")
        close(s[k], o["samples"], what=f"chain {k} draws")
        close(d[k], o["diagnostics"], what=f"chain {k} diagnostics")


def test_cfg4_dense1000_1024_chains(wa):
    """SURVEY 8d cfg 4, one GPU's shard: D = 1000 dense Gaussian (kappa = 100), 1024 chains, fixedLeapFrog, M = 10."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import make_dense_problem
    C, D, IT = 1024, 1000, 2
    A, q0 = make_dense_problem(D, 100.0, C, SEED)
    e = wa.Engine(D, C, integrator=ol.I_FIXED, H0=0.5, M=10, generated=ol.G_Q0_SUMSQ).set_target(ol.T_DENSEGAUSS, A)
    assert "split" in e.kernel_name
    s, d = e.run(q0, SEED, IT)
    s = s.transpose(1, 2).cpu().numpy(); d = d.cpu().numpy(); e.close()
    assert np.isfinite(s).all() and np.isfinite(d).all()
    steps = d[:, :, 6] + d[:, :, 7]
    np.testing.assert_array_equal(steps, 2.0 ** d[:, :, 20] - 1)
    assert (d[:, :, 17] < 5.0).all()
    for k in (0, 1, 63, 64, 511, 777, 1000, 1023):
        o = ol.walnuts(ol.T_DENSEGAUSS, q0[k], integrator=ol.I_FIXED, tpar=A.ravel(), H0=0.5, numIter=IT, M=10, seed=SEED, chain=k,
                       generated=ol.G_Q0_SUMSQ)
        np.testing.assert_array_equal(d[k][:, INT_COLS], o["diagnostics"][:, INT_COLS], err_msg=f"chain {k}")
        close(s[k], o["samples"], what=f"chain {k} draws")
        close(d[k], o["diagnostics"], what=f"chain {k} diagnostics")


@pytest.mark.parametrize("name,integ,H0,minC", [("D", ol.I_ADAPT_D, 0.1, 3), ("R2P", ol.I_ADAPT_R2P, 0.1, 3), ("fixed", ol.I_FIXED, 0.002, 0)])
def test_cfg5_stockwatson_256_chains(wa, name, integ, H0, minC):
    """SURVEY 8d cfg 5, one GPU's shard: StockWatson (D = 756), 256 chains, M = 14, q0 = 0 -- the three runs of mainSW.py:41-80
    (adaptLeapFrogD / adaptLeapFrogR2P with minC = 3, H0 = 0.1, delta0 = 0.3; fixedLeapFrog with H0 = 0.002); chains differ only by their
    random streams.  Round 5: these plain sampling runs are wn_k_orbit2.hip's (trajectory, bookkeeping and -- adaptive integrators --
    two helper waves that run the next refinement levels of a search, per chain)."""
    sw = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stockwatson_data.npz"))
    T = int(sw["T"][0]); D = 3 * T; C, IT = 256, 2
    tpar = np.concatenate([[float(T)], sw["y"]])
    q0 = np.zeros((C, D))
    e = wa.Engine(D, C, integrator=integ, H0=H0, delta0=0.3, M=14, minC=minC, generated=ol.G_FIRSTK, gen_k=2).set_target(ol.T_STOCKWATSON, tpar)
    assert "wn_orbit2_kernel<stockwatson" in e.kernel_name, e.kernel_name
    s, d = e.run(q0, SEED, IT)
    s = s.transpose(1, 2).cpu().numpy(); d = d.cpu().numpy(); e.close()
    assert np.isfinite(s).all()
    ok = np.isin(d[:, :, 19], [0, 4, -4])
    if integ != ol.I_FIXED:
        assert (d[:, :, 21][ok] >= 3).all() and (d[:, :, 8][ok] >= 3).all()  # every accepted refinement level is >= minC
        assert (d[:, :, 6][ok] >= 8 * (2.0 ** d[:, :, 20][ok] - 1)).all()     # >= 2^minC micro steps per macro step
    for k in (0, 100, 200, 255):
        o = ol.walnuts(ol.T_STOCKWATSON, q0[k], integrator=integ, tpar=tpar, H0=H0, delta0=0.3, numIter=IT, M=14, minC=minC,
                       seed=SEED, chain=k, generated=ol.G_FIRSTK, gen_k=2)
        np.testing.assert_array_equal(d[k][:, INT_COLS], o["diagnostics"][:, INT_COLS], err_msg=f"chain {k}")
        close(s[k], o["samples"], what=f"chain {k} draws")
        close(d[k], o["diagnostics"], what=f"chain {k} diagnostics")


@pytest.mark.parametrize("D,C", [(1000, 256), (600, 384)])
def test_split_mode_in_two_parts_is_the_same_run(wa, D, C):
    """Split mode cut into two sub-batches, each on a stream and a host thread of its own (wn_config.split_parts = 2; opt-in):
    the chains are independent and keyed by their global index, so the draws are the one-part run's bit for bit; with the k range of
    a GEMM tile split over two workgroups that add into zeroed rows (split_ks = 2) the gradient's sum is associated differently:
    integers identical, reals to 1e-9."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import make_dense_problem
    IT = 3
    A, q0 = make_dense_problem(D, 100.0, C, SEED)

    def go(parts, ks):
        e = wa.Engine(D, C, integrator=ol.I_FIXED, H0=0.5, M=8, generated=ol.G_IDENTITY, split_parts=parts, split_ks=ks).set_target(ol.T_DENSEGAUSS, A)
        assert "split" in e.kernel_name
        s, d = e.run(q0, SEED, IT)
        s = s.cpu().numpy(); d = d.cpu().numpy(); e.close()
        return s, d

    s1, d1 = go(1, 1)
    s2, d2 = go(2, 1)
    np.testing.assert_array_equal(s2, s1)
    np.testing.assert_array_equal(d2, d1)
    s3, d3 = go(2, 2)
    np.testing.assert_array_equal(d3[:, :, INT_COLS], d1[:, :, INT_COLS])
    close(s3, s1, rtol=1e-9, atol=1e-10, what="draws, k split over two workgroups")
    s4, d4 = go(1, 1)  # (and the one-part run after them: the zeroed exchange rows and the side streams leave nothing behind)
    np.testing.assert_array_equal(s4, s1)
