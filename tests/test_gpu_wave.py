"""GPU parity tests of the one-chain-per-wavefront layout (wn_policy_wave.h):
  * persistent kernels for large std / diagonal Gaussians (256 < D <= 1024),
  * split mode with the batched FP64-MFMA gradient kernel (dense Gaussian, 128 < D <= 1024,
    SURVEY.md 8d cfg 4),
  * split mode with a caller-evaluated target (WN_T_EXTERNAL): the reference's lpFun plugin as a
    batched device callable or as a per-chain host callable.
Same bar as test_gpu_parity.py: per-step (q, v) trajectories against the CPU oracle on the same
seed, rtol 1e-6, integers exact."""
import numpy as np
import pytest

import oracle_lib as ol
from test_gpu_parity import SEED, close, compare_chain, dense_problem, run_engine, wa  # noqa: F401

pytestmark = pytest.mark.gpu

WAVE = [
    # name, target, D, integrator, kwargs, C, check chains
    ("std300_fixed_wave", ol.T_STDGAUSS, 300, ol.I_FIXED, dict(H0=1.4 * 300 ** -0.25, M=7, numIter=4), 5, (0, 4)),
    ("diag600_R2P_wave", ol.T_DIAGGAUSS, 600, ol.I_ADAPT_R2P, dict(H0=0.9, delta0=0.3, M=6, numIter=3), 3, (0, 2)),
    ("diag1000_D_wave", ol.T_DIAGGAUSS, 1000, ol.I_ADAPT_D, dict(H0=0.9, delta0=0.3, M=5, numIter=3), 2, (1,)),
    ("std1024_fixed_wave", ol.T_STDGAUSS, 1024, ol.I_FIXED, dict(H0=1.4 * 1024 ** -0.25, M=6, numIter=3), 2, (0, 1)),
    ("diag400_yoshida_wave", ol.T_DIAGGAUSS, 400, ol.I_ADAPT_YOSHIDA, dict(H0=1.2, delta0=0.1, M=5, numIter=3), 2, (0,)),
    ("dense200_fixed_split", ol.T_DENSEGAUSS, 200, ol.I_FIXED, dict(H0=0.5, M=6, numIter=3), 70, (0, 63, 64, 69)),
    ("dense300_R2P_split", ol.T_DENSEGAUSS, 300, ol.I_ADAPT_R2P, dict(H0=1.2, delta0=0.3, M=5, numIter=3), 6, (0, 5)),
    ("dense520_D_split", ol.T_DENSEGAUSS, 520, ol.I_ADAPT_D, dict(H0=1.2, delta0=0.3, M=4, numIter=2), 3, (2,)),
    ("dense1000_fixed_split", ol.T_DENSEGAUSS, 1000, ol.I_FIXED, dict(H0=0.5, M=5, numIter=2), 5, (0, 4)),
    # adaptLeapFrogFlowD (adaptiveIntegrators.py:246-356): Gaussians of up to 128 dimensions run in the cooperative MFMA layout (round 5: a
    # micro step is two ticks of the group's machine -- std 60-D 3x, dense 40-D 20x the layouts below, tools/flowd_ab.py); beyond it one
    # chain per wavefront, dense ones in split mode (two exchanges per micro step)
    ("std60_F_coop", ol.T_STDGAUSS, 60, ol.I_ADAPT_FLOWD, dict(H0=1.2, delta0=0.02, M=5, numIter=3), 3, (0, 2)),
    ("diag100_F_coop", ol.T_DIAGGAUSS, 100, ol.I_ADAPT_FLOWD, dict(H0=0.9, delta0=0.05, M=5, numIter=3), 20, (1, 19)),
    ("diag300_F_wave", ol.T_DIAGGAUSS, 300, ol.I_ADAPT_FLOWD, dict(H0=0.9, delta0=0.05, M=5, numIter=3), 3, (1,)),
    ("dense40_F_coop", ol.T_DENSEGAUSS, 40, ol.I_ADAPT_FLOWD, dict(H0=1.0, delta0=0.05, M=5, numIter=3), 5, (0, 4)),
    ("dense100_F_coop", ol.T_DENSEGAUSS, 100, ol.I_ADAPT_FLOWD, dict(H0=1.0, delta0=0.05, M=5, numIter=3), 37, (0, 16, 36)),
    ("dense128_F_coop", ol.T_DENSEGAUSS, 128, ol.I_ADAPT_FLOWD, dict(H0=1.0, delta0=0.05, M=4, numIter=2), 17, (3, 16)),
    ("dense200_F_split", ol.T_DENSEGAUSS, 200, ol.I_ADAPT_FLOWD, dict(H0=1.0, delta0=0.05, M=4, numIter=2), 3, (2,)),
]


@pytest.mark.parametrize("case", WAVE, ids=[b[0] for b in WAVE])
def test_wave_layout_matches_oracle(wa, case):
    name, tk, D, ig, kw, C, check = case
    rng = np.random.default_rng(43)
    tpar = None
    if tk == ol.T_DENSEGAUSS:
        tpar, q0 = dense_problem(D, 100.0 if D >= 1000 else 1e4, C, 6)
        tpar = tpar.ravel()
    elif tk == ol.T_DIAGGAUSS:
        tpar = 1.0 / np.logspace(0, 3, D)
        q0 = rng.normal(size=(C, D)) / np.sqrt(tpar)
    else:
        q0 = rng.normal(size=(C, D))
    c = dict(target=tk, integrator=ig, H0=kw["H0"], s=0.2, delta0=kw.get("delta0", 0.05), M=kw["M"],
             numIter=kw["numIter"], minC=0, maxC=10, R2Pprob0=2.0 / 3.0, generated=0, gen_k=0, tpar=tpar)
    cap = 3 * (2 ** kw["M"]) * kw["numIter"] // 2
    out = run_engine(wa, c, q0, cap, chain_offset=9000)
    assert name.rsplit("_", 1)[1] in out["kernel"], (name, out["kernel"])  # the layout the case is named after
    for k in check:
        o = ol.walnuts(tk, q0[k], integrator=ig, tpar=tpar, H0=c["H0"], delta0=c["delta0"],
                       numIter=c["numIter"], M=c["M"], seed=SEED, chain=9000 + k, trace_cap=cap)
        assert o["ncalls"] <= cap
        compare_chain(out, k, o["trace"], o["samples"], o["diagnostics"], f"{name} chain {k} [{out['kernel']}]")


def test_wave_warmup_and_generated(wa):
    """Warm-up adaptation and generated = [q0, sum q^2] in the wave layout."""
    rng = np.random.default_rng(8)
    D, C = 320, 3
    q0 = rng.normal(size=(C, D))
    c = dict(target=ol.T_STDGAUSS, integrator=ol.I_ADAPT_R2P, H0=0.5, s=0.2, delta0=0.1, M=6, numIter=24,
             minC=0, maxC=10, R2Pprob0=2.0 / 3.0, generated=ol.G_Q0_SUMSQ, gen_k=0, tpar=None, warmupIter=16)
    out = run_engine(wa, c, q0, 4, chain_offset=50)
    # (the engine's kernel name is that of its plain sampling runs -- round 5: the trajectory-first kernel's wide mode; this run is traced:
    #  the one-stream engine of the wave layout)
    assert "wave" in out["kernel"] or "wn_orbit_kernel<diag,NE=8" in out["kernel"]
    for k in (0, 2):
        o = ol.walnuts(ol.T_STDGAUSS, q0[k], integrator=ol.I_ADAPT_R2P, H0=0.5, delta0=0.1, numIter=24, M=6, seed=SEED,
                       chain=50 + k, warmupIter=16, generated=ol.G_Q0_SUMSQ)
        close(out["diagnostics"][k][:, [15, 18]], o["diagnostics"][:, [15, 18]], what=f"H/delta chain {k}")
        np.testing.assert_array_equal(out["diagnostics"][k][:, [6, 7, 19, 20]], o["diagnostics"][:, [6, 7, 19, 20]])
        close(out["samples"][k], o["samples"], what=f"samples chain {k}")


def test_split_warmup_generated_and_orbit_stats(wa):
    """Split mode (dense 200-D) with warm-up adaptation, generated = [q0, sum q^2] (a deferred reduction that is
    parked between launches) and recordOrbitStats, chain by chain against the oracle."""
    D, C = 200, 4
    A, q0 = dense_problem(D, 1e3, C, 21)
    c = dict(target=ol.T_DENSEGAUSS, integrator=ol.I_ADAPT_D, H0=0.6, s=0.2, delta0=0.2, M=5, numIter=16,
             minC=0, maxC=10, R2Pprob0=2.0 / 3.0, generated=ol.G_Q0_SUMSQ, gen_k=0, tpar=A.ravel(), warmupIter=12)
    out = run_engine(wa, c, q0, 4, chain_offset=60)
    assert "split" in out["kernel"]
    for k in (0, 3):
        o = ol.walnuts(ol.T_DENSEGAUSS, q0[k], integrator=ol.I_ADAPT_D, tpar=A.ravel(), H0=0.6, delta0=0.2, numIter=16, M=5,
                       seed=SEED, chain=60 + k, warmupIter=12, generated=ol.G_Q0_SUMSQ)
        close(out["diagnostics"][k][:, [15, 18]], o["diagnostics"][:, [15, 18]], what=f"H/delta chain {k}")
        np.testing.assert_array_equal(out["diagnostics"][k][:, [6, 7, 19, 20]], o["diagnostics"][:, [6, 7, 19, 20]])
        close(out["samples"][k], o["samples"], what=f"samples chain {k}")
    e = wa.Engine(D, C, integrator=ol.I_FIXED, H0=0.5, M=5, generated=ol.G_FIRSTK, gen_k=3, chain_offset=60)
    e.set_target(ol.T_DENSEGAUSS, A.ravel())
    s, d, omin, omax = e.run_orbit_stats(q0, SEED, 4)
    omin = omin.cpu().numpy(); omax = omax.cpu().numpy(); e.close()
    o = ol.walnuts(ol.T_DENSEGAUSS, q0[2], integrator=ol.I_FIXED, tpar=A.ravel(), H0=0.5, numIter=4, M=5, seed=SEED,
                   chain=62, generated=ol.G_FIRSTK, gen_k=3, recordOrbitStats=True)
    close(omin[2].T, o["orbitMin"], what="orbitMin split")
    close(omax[2].T, o["orbitMax"], what="orbitMax split")


def test_split_dense_full_size_properties(wa):
    """SURVEY.md 8d cfg 4 shape (D = 1000, kappa = 100, fixedLeapFrog, M = 10) at a reduced chain
    count: size-independent properties -- every chain finishes, step counts are consistent with the
    tree depths, energy errors are bounded, sharding the chains does not change any draw."""
    D, C = 1000, 96
    A, q0 = dense_problem(D, 100.0, C, 11)
    def run(qs, off):
        e = wa.Engine(D, qs.shape[0], integrator=ol.I_FIXED, H0=0.5, M=10, generated=ol.G_Q0_SUMSQ, chain_offset=off)
        e.set_target(ol.T_DENSEGAUSS, A.ravel())
        s, d = e.run(qs, SEED, 3)
        s = s.cpu().numpy(); d = d.cpu().numpy(); ms = e.last_run_ms(); e.close()
        return s, d, ms
    s, d, ms = run(q0, 0)
    assert np.isfinite(s).all() and np.isfinite(d).all()
    steps = d[:, :, 6] + d[:, :, 7]
    assert (steps >= 1).all() and (steps <= 2 ** 10 - 1).all()
    np.testing.assert_array_equal(steps, 2.0 ** d[:, :, 20] - 1)  # fixedLeapFrog: 2^NdoublComputed - 1 gradient calls
    assert (d[:, :, 17] < 5.0).all()  # orbit energy error stays bounded (stable step size)
    s2, d2, _ = run(q0[64:], 64)
    np.testing.assert_array_equal(s2, s[64:])
    np.testing.assert_array_equal(d2, d[64:])


def test_external_batched_callable_matches_fused_target(wa):
    """WN_T_EXTERNAL with a batched torch callable for stdGauss reproduces the fused kernel."""
    import torch
    from walnuts_amd import adaptiveIntegrators as ai, targetDistr as td
    rng = np.random.default_rng(3)
    q0 = rng.normal(size=(9, 10))
    kw = dict(integrator=ai.adaptLeapFrogR2P, H0=0.9, delta0=0.3, numIter=12, warmupIter=0, M=7, seed=SEED, chain_offset=40)
    s0, d0 = wa.WALNUTS(td.stdGauss, q0, **kw)
    calls = [0]
    def fn(q):
        calls[0] += 1
        assert q.is_cuda and q.shape == (9, 10)
        return -0.5 * (q * q).sum(dim=1), -q
    s1, d1 = wa.WALNUTS(td.batched(fn, dim=10), q0, **kw)
    assert calls[0] >= (d0[:, :, 6] + d0[:, :, 7]).sum(axis=1).max()
    close(s1, s0, what="samples external vs fused")
    np.testing.assert_array_equal(d1[:, :, [6, 7, 19, 20]], d0[:, :, [6, 7, 19, 20]])


def test_funnel10rescaled_batched_target(wa):
    """targetDistr.funnel10rescaled (targetDistr.py:81-86) is a batched device callable: the same chain through
    the host-callable path with the formula written in NumPy gives the same draws."""
    from walnuts_amd import adaptiveIntegrators as ai, targetDistr as td
    S = np.ones(11); S[0] = 3.0
    def ref(q, hessian=False):
        lp, g = ol.lpfun(ol.T_FUNNEL10, S * q)
        return [lp, S * g]
    q0 = np.random.default_rng(6).normal(size=(2, 11)) * 0.3
    kw = dict(integrator=ai.adaptLeapFrogD, H0=0.3, delta0=0.1, numIter=6, warmupIter=0, M=6, seed=9)
    s1, d1 = wa.WALNUTS(td.funnel10rescaled, q0, **kw)
    s2, d2 = wa.WALNUTS(ref, q0, **kw)
    close(s1, s2, what="funnel10rescaled torch vs numpy callable")
    np.testing.assert_array_equal(d1[:, :, [6, 7, 19, 20]], d2[:, :, [6, 7, 19, 20]])


def test_external_host_callable_is_the_reference_plugin_contract(wa):
    """A plain per-chain callable lpFun(q) -> [lp, grad] (targetDistr.py:18) drives the sampler: with
    the oracle's funnel10 as that callable the chains must reproduce the oracle's own run."""
    from walnuts_amd import adaptiveIntegrators as ai
    rng = np.random.default_rng(4)
    C = 3
    q0 = np.zeros((C, 11)); q0[:, 0] = rng.normal(size=C)
    q0[:, 1:] = np.exp(0.5 * q0[:, :1]) * rng.normal(size=(C, 10))
    def lpFun(q, hessian=False):
        lp, g = ol.lpfun(ol.T_FUNNEL10, q)
        return [lp, g]
    s, d = wa.WALNUTS(lpFun, q0, integrator=ai.adaptLeapFrogR2P, H0=0.3, delta0=0.3, numIter=5, warmupIter=0, M=7,
                      seed=SEED, chain_offset=77)
    for k in range(C):
        o = ol.walnuts(ol.T_FUNNEL10, q0[k], integrator=ol.I_ADAPT_R2P, H0=0.3, delta0=0.3, numIter=5, M=7, seed=SEED,
                       chain=77 + k)
        close(s[k], o["samples"], what=f"host-callable funnel chain {k}")
        np.testing.assert_array_equal(d[k][:, [6, 7, 19, 20]], o["diagnostics"][:, [6, 7, 19, 20]])
        close(d[k], o["diagnostics"], what=f"diagnostics chain {k}")
    # NaN from the callable is a per-chain event (mainSW.py:24-26), never an abort
    def bad(q, hessian=False):
        return [float("nan"), np.zeros_like(q)] if q[0] > 1e300 else lpFun(q)
    wa.WALNUTS(bad, q0[0], integrator=ai.fixedLeapFrog, H0=0.3, numIter=2, warmupIter=0, M=4, seed=1)


# ---- StockWatson (BASELINE.json config 5) -----------------------------------------------------------
import os  # noqa: E402

_SW = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stockwatson_data.npz"))
SW_T = int(_SW["T"][0])
SW_TPAR = np.concatenate([[float(SW_T)], _SW["y"]])


def test_stockwatson_lpfun_through_c_abi(wa):
    """wn_lpfun (scalar device form) against the oracle at random points."""
    rng = np.random.default_rng(0)
    B, D = 5, 3 * SW_T
    q = 0.4 * rng.normal(size=(B, D))
    e = wa.Engine(D, B).set_target(ol.T_STOCKWATSON, SW_TPAR)
    lp, g = e.lpfun(q)
    lp = lp.cpu().numpy(); g = g.cpu().numpy(); e.close()
    for b in range(B):
        rl, rg = ol.lpfun(ol.T_STOCKWATSON, q[b], SW_TPAR)
        close(lp[b], rl, rtol=1e-10, what="SW lp")
        close(g[b], rg, rtol=1e-9, atol=1e-9, what="SW grad")


@pytest.mark.parametrize("ig,kw", [(ol.I_FIXED, dict(H0=0.002, M=6, numIter=3)),
                                   (ol.I_ADAPT_D, dict(H0=0.1, delta0=0.3, M=5, numIter=3, minC=3)),
                                   (ol.I_ADAPT_R2P, dict(H0=0.1, delta0=0.3, M=5, numIter=3, minC=3)),
                                   (ol.I_ADAPT_FLOWD, dict(H0=0.05, delta0=0.01, M=4, numIter=2))],
                         ids=["fixed", "adaptD_minC3", "R2P_minC3", "flowD"])
def test_stockwatson_matches_oracle(wa, ig, kw):
    """mainSW.py:41-80 settings on the fused wave kernel: per-step (q, v) against the oracle."""
    rng = np.random.default_rng(12)
    D, C = 3 * SW_T, 3
    q0 = 0.1 * rng.normal(size=(C, D))
    q0[0] = 0.0  # mainSW.py's initq.npy is not in the reference tree: q0 = 0 (SURVEY.md 8d cfg 5)
    c = dict(target=ol.T_STOCKWATSON, integrator=ig, H0=kw["H0"], s=0.2, delta0=kw.get("delta0", 0.05), M=kw["M"],
             numIter=kw["numIter"], minC=kw.get("minC", 0), maxC=10, R2Pprob0=2.0 / 3.0, generated=0, gen_k=0, tpar=SW_TPAR)
    cap = 40 * (2 ** kw["M"]) * kw["numIter"]
    out = run_engine(wa, c, q0, cap, chain_offset=500)
    assert "stockwatson" in out["kernel"]
    for k in (0, C - 1):
        o = ol.walnuts(ol.T_STOCKWATSON, q0[k], integrator=ig, tpar=SW_TPAR, H0=c["H0"], delta0=c["delta0"],
                       numIter=c["numIter"], M=c["M"], minC=c["minC"], seed=SEED, chain=500 + k, trace_cap=cap)
        assert o["ncalls"] <= cap
        compare_chain(out, k, o["trace"], o["samples"], o["diagnostics"], f"SW chain {k}")


@pytest.mark.parametrize("name", ["D", "R2P", "fixed"])
def test_stockwatson_engine_matches_reference_sampler_at_config5_size(wa, name):
    """The three runs of mainSW.py:41-80 at their size (D = 756, minC = 3, M = 14; the fixedLeapFrog run at M = 7) against
    the UNMODIFIED Python reference sampler driven by the NumPy restatement of sw_innov.stan
    (tests/golden/stockwatson_sampler.npz, made by tests/golden/make_golden_sw_sampler.py): every integrator call's
    If / Ib / c / nEval exactly, the stored (q, v) rows, draws and diagnostics to rtol 1e-6."""
    from test_stockwatson import sw_sampler_case
    kw, chain, seed, ref = sw_sampler_case(name)
    assert seed == SEED
    ncall = len(ref["trace_scalars"])
    c = dict(target=ol.T_STOCKWATSON, integrator=kw["integrator"], H0=kw["H0"], s=0.2, delta0=kw["delta0"], M=kw["M"],
             numIter=kw["numIter"], minC=kw["minC"], maxC=10, R2Pprob0=2.0 / 3.0, generated=0, gen_k=0, tpar=SW_TPAR)
    out = run_engine(wa, c, np.zeros((1, 3 * SW_T)), ncall + 8, chain_offset=chain)
    assert "stockwatson" in out["kernel"]
    assert out["trace_n"][0] == ncall
    tr = out["trace"][0, :ncall]
    for col in (0, 2, 6, 7, 8, 10, 11):
        np.testing.assert_array_equal(tr[:, col], ref["trace_scalars"][:, col], err_msg=f"trace col {col}")
    close(tr[:, [3, 4, 5, 9]], ref["trace_scalars"][:, [3, 4, 5, 9]], what="h / Ham0 / lp / lwt")
    close(tr[ref["trace_rows"], 12:], ref["trace_qv"], what="(q, v) of the stored integrator calls")
    d = out["diagnostics"][0]
    np.testing.assert_array_equal(d[:, [6, 7, 8, 9, 19, 20, 21, 22]], ref["diagnostics"][:, [6, 7, 8, 9, 19, 20, 21, 22]])
    close(d, ref["diagnostics"], what="diagnostics")
    close(out["samples"][0], ref["samples"], what="samples")


def test_stockwatson_python_api(wa):
    from walnuts_amd import adaptiveIntegrators as ai, targetDistr as td
    s, d = wa.WALNUTS(td.stockWatson(_SW["y"]), np.zeros((2, 3 * SW_T)), integrator=ai.adaptLeapFrogD, H0=0.1, delta0=0.3,
                      numIter=4, warmupIter=0, M=6, igrAux=ai.integratorAuxPar(minC=3), seed=3)
    assert s.shape == (2, 3 * SW_T, 5) and np.isfinite(s).all() and (d[:, :, 21] >= 3).all()
    # generated = param_constrain(q, include_tp=True) (mainSW.py:28-29): 3T parameters + sigma, gpri, z, x, tau
    s2, d2 = wa.WALNUTS(td.stockWatson(_SW["y"]), np.zeros(3 * SW_T), generated=td.stockWatsonConstrain(_SW["y"]),
                        integrator=ai.adaptLeapFrogD, H0=0.1, delta0=0.3, numIter=3, warmupIter=0, M=5,
                        igrAux=ai.integratorAuxPar(minC=3), seed=3)
    T = SW_T
    assert s2.shape == (3 * T + 2 + (T - 1) + 2 * T, 4)
    q = s2[:3 * T, 3]
    np.testing.assert_allclose(s2[3 * T, 3], np.exp(-0.5 * q[0]))
    x = q[T] + np.concatenate([[0.0], np.cumsum(np.exp(-0.5 * q[0]) * q[T + 1:2 * T])])
    np.testing.assert_allclose(s2[3 * T + 2 + (T - 1):3 * T + 2 + (T - 1) + T, 3], x, rtol=1e-12, atol=1e-12)


def test_stockwatson_device_generated_matches_host_form(wa):
    """generated = param_constrain(q, include_tp=True) (mainSW.py:28-29, sw_innov.stan:17-35) on the device
    (WN_G_SW_CONSTRAIN) against the host form of the same transform applied to the stored identity draws: 1e-12."""
    from walnuts_amd import adaptiveIntegrators as ai, targetDistr as td
    T = SW_T
    q0 = 0.05 * np.random.default_rng(2).normal(size=(3, 3 * T))
    kw = dict(integrator=ai.adaptLeapFrogD, H0=0.1, delta0=0.3, numIter=4, warmupIter=0, M=5, igrAux=ai.integratorAuxPar(minC=3), seed=3)
    sd, dd = wa.WALNUTS(td.stockWatson(_SW["y"]), q0, generated=wa.sw_constrain, **kw)
    sh, dh = wa.WALNUTS(td.stockWatson(_SW["y"]), q0, generated=td.stockWatsonConstrain(_SW["y"]), **kw)
    assert sd.shape == sh.shape == (3, 6 * T + 1, 5)
    np.testing.assert_array_equal(dd, dh)
    np.testing.assert_array_equal(sd[:, :3 * T], sh[:, :3 * T])
    np.testing.assert_allclose(sd, sh, rtol=1e-12, atol=1e-13)
    with pytest.raises(ValueError):
        wa.WALNUTS(td.stdGauss, np.zeros(9), generated=wa.sw_constrain, numIter=1, warmupIter=0)


@pytest.mark.parametrize("tk,D,ig,minC", [(ol.T_FUNNEL10, 11, ol.I_ADAPT_R2P, 0), (ol.T_FUNNEL10, 11, ol.I_ADAPT_D, 0),
                                           (ol.T_FUNNEL10, 11, ol.I_ADAPT_R2P, 2), (ol.T_CORRGAUSS, 2, ol.I_ADAPT_R2P, 0),
                                           (ol.T_STOCKWATSON, 0, ol.I_ADAPT_D, 3)])
def test_nested_driver_is_bit_identical_to_the_tick_loop(wa, tk, D, ig, minC):
    """Plain sampling runs of the one-chain-per-wave kernels go through Engine::run_nested() (nested loops, the
    integrator's searches written out, specialised engine); a traced run of the same chains goes through the generic
    tick loop Engine::run().  Same member functions, same reductions, same draw order: every draw and every
    diagnostics column must agree bit for bit (integer columns and floats alike)."""
    import torch
    rng = np.random.default_rng(5)
    tpar = None
    C = 37
    if tk == ol.T_STOCKWATSON:
        sw = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stockwatson_data.npz"))
        T = int(sw["T"][0]); D = 3 * T
        tpar = np.concatenate([[float(T)], sw["y"]])
        q0 = np.zeros((6, D)); C = 6
        kw = dict(H0=0.1, delta0=0.3, M=8, numIter=3)
    elif tk == ol.T_FUNNEL10:
        q0 = np.zeros((C, 11)); q0[:, 0] = 2.0 * rng.normal(size=C); q0[:, 1:] = np.exp(0.5 * q0[:, :1]) * rng.normal(size=(C, 10))
        kw = dict(H0=0.3, delta0=0.3, M=9, numIter=12)
    else:
        q0 = rng.normal(size=(C, D))
        kw = dict(H0=0.4, delta0=0.1, M=7, numIter=10)
    outs = []
    for traced in (False, True):
        e = wa.Engine(D, C, integrator=ig, H0=kw["H0"], delta0=kw["delta0"], M=kw["M"], minC=minC, chain_offset=123)
        e.set_target(tk, tpar)
        assert "wave" in e.kernel_name or "orbit" in e.kernel_name, e.kernel_name
        r = e.run(q0, SEED, kw["numIter"], trace_cap=4 if traced else 0)
        outs.append((r[0].cpu().numpy(), r[1].cpu().numpy()))
        orbit = "orbit" in e.kernel_name
        e.close()
    if orbit:
        # <= 16-D targets: the plain run is the trajectory-first kernel (wn_k_orbit.hip), the traced run the tick loop.  Same
        # arithmetic per micro step and per sum; the orbit energy error of the FIRST iteration alone may differ in the last bit
        # (its H0 comes from lp(q0), evaluated at another inlined code site: tests/test_gpu_orbit.py same_run)
        k = [c for c in range(24) if c != 17]
        np.testing.assert_array_equal(outs[0][1][:, :, k], outs[1][1][:, :, k], err_msg="diagnostics")
        np.testing.assert_allclose(outs[0][1][:, :, 17], outs[1][1][:, :, 17], rtol=1e-12, atol=1e-13)
    else:
        np.testing.assert_array_equal(outs[0][1], outs[1][1], err_msg="diagnostics")
    np.testing.assert_array_equal(outs[0][0], outs[1][0], err_msg="draws")
    assert outs[0][1][:, :, 6].sum() > 0


@pytest.mark.parametrize("T,ig,kw", [(SW_T, ol.I_FIXED, dict(H0=0.01, M=5, numIter=3)),
                                      (SW_T, ol.I_ADAPT_D, dict(H0=0.1, delta0=0.3, M=4, numIter=3, minC=3)),
                                      (30, ol.I_ADAPT_R2P, dict(H0=0.2, delta0=0.3, M=5, numIter=4, minC=0))],
                         ids=["T252_fixed", "T252_adaptD_minC3", "T30_R2P"])
def test_stockwatson_orbit_statistics_with_param_constrain(wa, T, ig, kw):
    """Round 5 (VERDICT round 4 "missing" 5, ADVICE): recordOrbitStats=True with generated = StockWatson's param_constrain
    (WALNUTS.py:182-184,331-333 with mainSW.py:28-29; sw_innov.stan:17-35).  The wave kernel applies the transform to EVERY orbit state
    (WTgtStockWatson::constrain: the series from the gradient's own forward sums) -- round 4 returned WN_E_UNSUPPORTED.  Against the
    oracle, whose generated() restates the Stan block in its own (sequential) order: per-iteration minima / maxima of all 6T+1 entries,
    the draws, and min <= draw <= max up to the rounding of two summation orders."""
    from walnuts_amd import adaptiveIntegrators as ai, targetDistr as td
    y = _SW["y"][:T]
    tpar = np.concatenate([[float(T)], y])
    C = 3
    q0 = 0.05 * np.random.default_rng(T).normal(size=(C, 3 * T))
    integ = {ol.I_FIXED: ai.fixedLeapFrog, ol.I_ADAPT_D: ai.adaptLeapFrogD, ol.I_ADAPT_R2P: ai.adaptLeapFrogR2P}[ig]
    s, d, omin, omax = wa.WALNUTS(td.stockWatson(y), q0, generated=wa.sw_constrain, integrator=integ, H0=kw["H0"], delta0=kw.get("delta0", 0.05),
                                  numIter=kw["numIter"], warmupIter=0, M=kw["M"], igrAux=ai.integratorAuxPar(minC=kw.get("minC", 0)),
                                  recordOrbitStats=True, seed=SEED, chain_offset=70)
    assert s.shape == (C, 6 * T + 1, kw["numIter"] + 1) and omin.shape == (C, 6 * T + 1, kw["numIter"])
    for k in range(C):
        o = ol.walnuts(ol.T_STOCKWATSON, q0[k], integrator=ig, tpar=tpar, H0=kw["H0"], delta0=kw.get("delta0", 0.05), numIter=kw["numIter"],
                       M=kw["M"], minC=kw.get("minC", 0), seed=SEED, chain=70 + k, generated=ol.G_SW_CONSTRAIN, recordOrbitStats=True)
        np.testing.assert_array_equal(d[k][:, [6, 7, 19, 20]], o["diagnostics"][:, [6, 7, 19, 20]])
        close(s[k], o["samples"], rtol=1e-6, atol=1e-9, what=f"draws chain {k}")
        close(omin[k], o["orbitMin"], rtol=1e-6, atol=1e-9, what=f"orbitMin chain {k}")
        close(omax[k], o["orbitMax"], rtol=1e-6, atol=1e-9, what=f"orbitMax chain {k}")
        tol = 1e-9 * (1.0 + np.abs(s[k][:, 1:]))
        assert (omin[k] <= s[k][:, 1:] + tol).all() and (omax[k] >= s[k][:, 1:] - tol).all()
        assert (omax[k][3 * T:] > omin[k][3 * T:]).mean() > 0.9  # the transformed entries did move along the orbits
