"""GPU parity tests: the HIP engine, called through the C ABI, against
  (1) the golden fixtures captured from the Python reference, and
  (2) the CPU oracle on the same seeded inputs (the engine's RNG stream is replayed by the
      oracle's independent Philox restatement).
Tolerance for floating point: rtol 1e-6 (BASELINE.json north_star), integers exact.
"""
import os

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TR = np.load(os.path.join(G, "walnuts_traces.npz"))
KA = np.load(os.path.join(G, "integrator_kats.npz"))
CASES = [str(x) for x in TR["case_names"]]
SEED = int(TR["seed"][0])
RTOL, ATOL = 1e-6, 1e-9
INT_DIAG = (0, 1, 4, 5, 6, 7, 8, 9, 12, 13, 19, 20, 21, 22)


@pytest.fixture(scope="module")
def wa():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import walnuts_amd
    return walnuts_amd


def close(a, b, rtol=RTOL, atol=ATOL, what=""):
    a = np.asarray(a, float); b = np.asarray(b, float)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    with np.errstate(all="ignore"):
        both_nan = np.isnan(a) & np.isnan(b)
        same_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))
        ok = both_nan | same_inf | (np.abs(a - b) <= atol + rtol * np.abs(b))
    assert ok.all(), f"{what}: {np.count_nonzero(~ok)} mismatches, first at {np.argwhere(~ok)[:4].tolist()}, " \
                     f"a={a[~ok][:4]}, b={b[~ok][:4]}"


def load_case(name):
    p = name + "__"
    cfg = TR[p + "cfg"]; par = TR[p + "par"]
    c = dict(target=int(cfg[0]), integrator=int(cfg[1]), M=int(cfg[2]), numIter=int(cfg[3]),
             minC=int(cfg[4]), maxC=int(cfg[5]), generated=int(cfg[6]), gen_k=int(cfg[7]),
             chain=int(cfg[8]), H0=float(par[0]), s=float(par[1]), delta0=float(par[2]),
             R2Pprob0=float(par[3]), q0=TR[p + "q0"],
             tpar=TR[p + "tpar"] if (p + "tpar") in TR.files else None)
    w = TR[p + "warm"] if (p + "warm") in TR.files else np.array([0, 1, 0.8, 1, 0.6, 0.9])
    c.update(warmupIter=int(w[0]), adaptH=bool(w[1]), adaptHtarget=float(w[2]), adaptDelta=bool(w[3]),
             adaptDeltaTarget=float(w[4]), adaptDeltaQuantile=float(w[5]))
    ref = {k: TR[p + k] for k in ("samples", "diagnostics", "trace", "n_u")}
    if (p + "orbitMin") in TR.files:
        ref["orbitMin"] = TR[p + "orbitMin"]; ref["orbitMax"] = TR[p + "orbitMax"]
    return c, ref


def run_engine(wa, c, q0, ncalls_hint, chain_offset):
    q0 = np.atleast_2d(q0)
    e = wa.Engine(q0.shape[1], q0.shape[0], integrator=c["integrator"], H0=c["H0"],
                  stepSizeRandScale=c["s"], delta0=c["delta0"], M=c["M"], minC=c["minC"],
                  maxC=c["maxC"], R2Pprob0=c["R2Pprob0"], generated=c["generated"], gen_k=c["gen_k"],
                  chain_offset=chain_offset, warmupIter=c.get("warmupIter", 0), adaptH=c.get("adaptH", True),
                  adaptHtarget=c.get("adaptHtarget", 0.8), adaptDelta=c.get("adaptDelta", True),
                  adaptDeltaTarget=c.get("adaptDeltaTarget", 0.6),
                  adaptDeltaQuantile=c.get("adaptDeltaQuantile", 0.9), waves_per_group=c.get("waves_per_group", 0))
    e.set_target(c["target"], c["tpar"])
    s, d, tr, tn = e.run(q0, SEED, c["numIter"], trace_cap=ncalls_hint)
    out = dict(samples=s.transpose(1, 2).cpu().numpy(), diagnostics=d.cpu().numpy(),
               trace=tr.cpu().numpy(), trace_n=tn.cpu().numpy(), kernel=e.kernel_name)
    e.close()
    return out


def compare_chain(out, k, ref_trace, ref_samples, ref_diag, what):
    n = len(ref_trace)
    assert out["trace_n"][k] == n, f"{what}: {out['trace_n'][k]} integrator calls, reference {n}"
    tr = out["trace"][k, :n]
    for col in (0, 2, 6, 7, 8, 10, 11):  # iter, xi, If, Ib, c, nEvalF, nEvalB
        np.testing.assert_array_equal(tr[:, col], ref_trace[:, col], err_msg=f"{what} trace col {col}")
    close(tr[:, 3:6], ref_trace[:, 3:6], what=what + " h/Ham0/lp")
    close(tr[:, 9], ref_trace[:, 9], what=what + " lwt")
    close(tr[:, 12:], ref_trace[:, 12:], what=what + " per-step (q,v)")
    close(out["samples"][k], ref_samples, what=what + " samples")
    d = out["diagnostics"][k]
    for col in INT_DIAG:
        np.testing.assert_array_equal(d[:, col], ref_diag[:, col], err_msg=f"{what} diag col {col}")
    close(d, ref_diag, what=what + " diagnostics")


@pytest.mark.parametrize("name", CASES)
def test_golden_trace_through_c_abi(wa, name):
    """Every integrator call of every reference fixture, reproduced by the HIP engine."""
    c, ref = load_case(name)
    out = run_engine(wa, c, c["q0"], len(ref["trace"]) + 8, c["chain"])
    compare_chain(out, 0, ref["trace"], ref["samples"], ref["diagnostics"], name)


@pytest.mark.parametrize("name", [n for n in CASES if "_orbit" in n])  # incl. generated = [q0, sum q^2] (..._orbit_ess)
def test_record_orbit_stats_matches_reference(wa, name):
    """recordOrbitStats=True (WALNUTS.py:182-184, 724-725): per-iteration min/max of generated(q)."""
    c, ref = load_case(name)
    e = wa.Engine(c["q0"].size, 1, integrator=c["integrator"], H0=c["H0"], stepSizeRandScale=c["s"], delta0=c["delta0"],
                  M=c["M"], minC=c["minC"], maxC=c["maxC"], R2Pprob0=c["R2Pprob0"], generated=c["generated"],
                  gen_k=c["gen_k"], chain_offset=c["chain"])
    e.set_target(c["target"], c["tpar"])
    s, d, omin, omax = e.run_orbit_stats(c["q0"][None], SEED, c["numIter"])
    close(omin.cpu().numpy()[0].T, ref["orbitMin"], what="orbitMin")
    close(omax.cpu().numpy()[0].T, ref["orbitMax"], what="orbitMax")
    close(s.cpu().numpy()[0].T, ref["samples"], what="samples")
    e.close()
    from walnuts_amd import adaptiveIntegrators as ai, targetDistr as td
    if name == "std3_fixed_orbit":  # the 4-tuple return of the Python mirror
        r = wa.WALNUTS(td.stdGauss, c["q0"], integrator=ai.fixedLeapFrog, H0=c["H0"], numIter=c["numIter"], warmupIter=0,
                       M=c["M"], recordOrbitStats=True, seed=SEED, chain_offset=c["chain"])
        assert len(r) == 4 and r[2].shape == (3, c["numIter"])
        close(r[2], ref["orbitMin"], what="orbitMin via WALNUTS()")


def test_rng_tape_matches_oracle(wa):
    e = wa.Engine(37, 5, chain_offset=1000)
    u, z = e.rng_tape(SEED, 3, 50)
    u = u.cpu().numpy(); z = z.cpu().numpy()
    for c in range(5):
        uo, zo = ol.rng_tape(SEED, 1000 + c, 3, 50, 37)
        np.testing.assert_array_equal(u[c], uo)          # integer path: bit exact
        np.testing.assert_allclose(z[c], zo, rtol=1e-14, atol=1e-15)  # libm vs ocml log/sincos
    e.close()


@pytest.mark.parametrize("i", range(int(KA["n"][0])))
def test_integrator_kat_through_c_abi(wa, i):
    p = f"k{i}__"
    tk, kind, d, xi, minC, maxC = [int(x) for x in KA[p + "meta"]]
    Ham0, h, delta, u = [float(x) for x in KA[p + "par"]]
    tpar = KA["dense6_A"] if tk == ol.T_DENSEGAUSS else None
    e = wa.Engine(d, 1, integrator=kind, delta0=delta, minC=minC, maxC=maxC)
    e.set_target(tk, tpar)
    q1, v1, g1, io, do = e.integrator_step(KA[p + "q"], KA[p + "v"], KA[p + "g"], [Ham0], [h], [xi], [u])
    np.testing.assert_array_equal(io.cpu().numpy()[0], KA[p + "io"])
    close(do.cpu().numpy()[0][:2], KA[p + "do"][:2], what="lp,lwt")
    close(do.cpu().numpy()[0][3], KA[p + "do"][2], rtol=1e-6, what="igrConst")
    close(q1.cpu().numpy()[0], KA[p + "q1"], what="q"); close(v1.cpu().numpy()[0], KA[p + "v1"], what="v")
    close(g1.cpu().numpy()[0], KA[p + "g1"], what="grad")
    e.close()


@pytest.mark.parametrize("tname", ["std3", "std100", "funnel", "dense6", "corr", "smile", "funnel1", "modfunnel"])
def test_lpfun_through_c_abi(wa, tname):
    tk, d = [int(x) for x in KA[f"lp__{tname}__kind"]]
    Q = KA[f"lp__{tname}__q"]
    tpar = KA["dense6_A"] if tk == ol.T_DENSEGAUSS else None
    e = wa.Engine(d, Q.shape[0]).set_target(tk, tpar)
    lp, g = e.lpfun(Q)
    close(lp.cpu().numpy(), KA[f"lp__{tname}__lp"], rtol=1e-12, what="lp")
    close(g.cpu().numpy(), KA[f"lp__{tname}__g"], rtol=1e-12, what="grad")
    e.close()


def dense_problem(D, kappa, C, seed):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.normal(size=(D, D)))
    lam = np.logspace(0, np.log10(kappa), D)
    A = (Q * (1.0 / lam)) @ Q.T
    A = 0.5 * (A + A.T)
    q0 = rng.normal(size=(C, D)) @ (Q * np.sqrt(lam)).T
    return A, q0


BATCH = [
    # name, target, D, integrator, kwargs, C, check chains
    ("dense100_fixed", ol.T_DENSEGAUSS, 100, ol.I_FIXED, dict(H0=0.5, M=7, numIter=4), 48, (0, 17, 47)),
    ("dense100_R2P", ol.T_DENSEGAUSS, 100, ol.I_ADAPT_R2P, dict(H0=1.2, delta0=0.3, M=5, numIter=3), 20, (0, 19)),
    ("dense100_D", ol.T_DENSEGAUSS, 100, ol.I_ADAPT_D, dict(H0=1.2, delta0=0.3, M=5, numIter=3), 16, (3, 15)),
    ("dense64_fixed", ol.T_DENSEGAUSS, 64, ol.I_FIXED, dict(H0=0.6, M=6, numIter=4), 33, (0, 32)),
    ("dense128_fixed", ol.T_DENSEGAUSS, 128, ol.I_FIXED, dict(H0=0.6, M=6, numIter=3), 17, (16,)),
    ("dense37_fixed", ol.T_DENSEGAUSS, 37, ol.I_FIXED, dict(H0=0.6, M=6, numIter=4), 16, (5,)),
    ("std100_R2P", ol.T_STDGAUSS, 100, ol.I_ADAPT_R2P, dict(H0=1.4 * 100 ** -0.25, delta0=0.3, M=8, numIter=6), 40, (0, 39)),
    ("diag200_fixed", ol.T_DIAGGAUSS, 200, ol.I_FIXED, dict(H0=0.7, M=6, numIter=4), 16, (9,)),
    ("std10_fixed", ol.T_STDGAUSS, 10, ol.I_FIXED, dict(H0=0.9, M=8, numIter=8), 130, (0, 64, 129)),
    ("funnel_R2P", ol.T_FUNNEL10, 11, ol.I_ADAPT_R2P, dict(H0=0.3, delta0=0.3, M=8, numIter=6), 100, (0, 63, 64, 99)),
    ("funnel_D", ol.T_FUNNEL10, 11, ol.I_ADAPT_D, dict(H0=0.3, delta0=0.3, M=8, numIter=6), 70, (1, 69)),
]


@pytest.mark.parametrize("case", BATCH, ids=[b[0] for b in BATCH])
def test_batch_matches_oracle(wa, case):
    """Many chains in one launch (ragged chain counts, both layouts): selected chains must
    reproduce the oracle's per-step (q, v) trajectory on the same seed."""
    name, tk, D, ig, kw, C, check = case
    rng = np.random.default_rng(42)
    tpar = None
    if tk == ol.T_DENSEGAUSS:
        tpar, q0 = dense_problem(D, 1e4 if D >= 64 else 100.0, C, 5)
        tpar = tpar.ravel()
    elif tk == ol.T_DIAGGAUSS:
        tpar = 1.0 / np.logspace(0, 3, D)
        q0 = rng.normal(size=(C, D)) / np.sqrt(tpar)
    elif tk == ol.T_FUNNEL10:
        q0 = np.zeros((C, 11)); q0[:, 0] = 3.0 * rng.normal(size=C)
        q0[:, 1:] = np.exp(0.5 * q0[:, :1]) * rng.normal(size=(C, 10))
    else:
        q0 = rng.normal(size=(C, D))
    c = dict(target=tk, integrator=ig, H0=kw["H0"], s=0.2, delta0=kw.get("delta0", 0.05), M=kw["M"],
             numIter=kw["numIter"], minC=0, maxC=10, R2Pprob0=2.0 / 3.0, generated=0, gen_k=0, tpar=tpar)
    cap = 3 * (2 ** kw["M"]) * kw["numIter"] // 2
    out = run_engine(wa, c, q0, cap, chain_offset=7000)
    for k in check:
        o = ol.walnuts(tk, q0[k], integrator=ig, tpar=tpar, H0=c["H0"], delta0=c["delta0"],
                       numIter=c["numIter"], M=c["M"], seed=SEED, chain=7000 + k, trace_cap=cap)
        assert o["ncalls"] <= cap
        compare_chain(out, k, o["trace"], o["samples"], o["diagnostics"], f"{name} chain {k} [{out['kernel']}]")


def test_warmup_batch_matches_oracle(wa):
    """Warm-up adaptation in a batched launch (coop and lane layouts): H (col 15) and delta (col 18)
    per iteration and the draws agree with the oracle chain by chain."""
    rng = np.random.default_rng(5)
    for tk, D, ig, C, kw in [(ol.T_STDGAUSS, 40, ol.I_ADAPT_R2P, 20, dict(H0=0.4, delta0=0.1, M=6)),
                             (ol.T_STDGAUSS, 6, ol.I_FIXED, 70, dict(H0=0.3, delta0=0.05, M=6)),
                             (ol.T_FUNNEL10, 11, ol.I_ADAPT_D, 66, dict(H0=0.3, delta0=0.3, M=7))]:
        q0 = rng.normal(size=(C, D))
        c = dict(target=tk, integrator=ig, H0=kw["H0"], s=0.2, delta0=kw["delta0"], M=kw["M"], numIter=30,
                 minC=0, maxC=10, R2Pprob0=2.0 / 3.0, generated=0, gen_k=0, tpar=None, warmupIter=20)
        out = run_engine(wa, c, q0, 4, chain_offset=300)
        for k in (0, C - 1):
            o = ol.walnuts(tk, q0[k], integrator=ig, H0=c["H0"], delta0=c["delta0"], numIter=30, M=c["M"],
                           seed=SEED, chain=300 + k, warmupIter=20)
            close(out["diagnostics"][k][:, [15, 18]], o["diagnostics"][:, [15, 18]], what=f"H/delta {tk} chain {k}")
            np.testing.assert_array_equal(out["diagnostics"][k][:, [6, 7, 19, 20]], o["diagnostics"][:, [6, 7, 19, 20]])
            close(out["samples"][k], o["samples"], what=f"samples {tk} chain {k}")


def test_sharding_invariance(wa):
    """per-chain results do not depend on how chains are split over engines/GPUs."""
    A, q0 = dense_problem(100, 1e4, 64, 11)
    c = dict(target=ol.T_DENSEGAUSS, integrator=ol.I_FIXED, H0=0.5, s=0.2, delta0=0.05, M=6, numIter=3,
             minC=0, maxC=10, R2Pprob0=2.0 / 3.0, generated=1, gen_k=0, tpar=A.ravel())
    whole = run_engine(wa, c, q0, 8, chain_offset=0)
    a = run_engine(wa, c, q0[:32], 8, chain_offset=0)
    b = run_engine(wa, c, q0[32:], 8, chain_offset=32)
    np.testing.assert_array_equal(whole["samples"], np.concatenate([a["samples"], b["samples"]]))
    np.testing.assert_array_equal(whole["diagnostics"], np.concatenate([a["diagnostics"], b["diagnostics"]]))


def test_time_reversal_of_integrators(wa):
    """Invariant the reference's commented scaffold checks (adaptiveIntegrators.py:769-789):
    integrating forward and then from (q', -v') returns to (q, -v) with If/Ib swapped."""
    A, q0 = dense_problem(100, 1e4, 32, 3)
    rng = np.random.default_rng(0)
    v = rng.normal(size=q0.shape)
    for kind in (ol.I_FIXED, ol.I_ADAPT_D):
        e = wa.Engine(100, 32, integrator=kind, delta0=0.05).set_target(ol.T_DENSEGAUSS, A)
        lp, g = e.lpfun(q0)
        H0 = (-lp.cpu().numpy() + 0.5 * (v * v).sum(1))
        h = np.full(32, 0.9); xi = np.ones(32, dtype=np.int32)
        q1, v1, g1, io, do = e.integrator_step(q0, v, g, H0, h, xi)
        q2, v2, g2, io2, do2 = e.integrator_step(q1, -v1, g1, do[:, 2], h, xi)
        io = io.cpu().numpy(); io2 = io2.cpu().numpy()
        # the return trip retraces the path only where both directions choose the same refinement
        # (If == Ib; otherwise the reference zero-weights the step, adaptiveIntegrators.py:136)
        same = io[:, 2] == io[:, 3]
        assert same.sum() >= 16
        np.testing.assert_allclose(q2.cpu().numpy()[same], q0[same], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(v2.cpu().numpy()[same], -v[same], rtol=1e-9, atol=1e-10)
        if kind == ol.I_ADAPT_D:
            np.testing.assert_array_equal(io2[:, 2], io[:, 3])  # the return trip's If is this trip's Ib
        e.close()


def test_full_size_properties(wa):
    """BASELINE configs[1] at full size (100-D kappa=1e4, 1024 chains, fixedLeapFrog, M=10):
    size-independent properties -- step accounting, energy-error sanity, stop codes, and the
    exact invariant  #steps = sum over completed doublings of 2**i."""
    A, q0 = dense_problem(100, 1e4, 1024, 20251121)
    e = wa.Engine(100, 1024, integrator=ol.I_FIXED, H0=0.5, M=10, generated=1).set_target(ol.T_DENSEGAUSS, A)
    s, d = e.run(q0, SEED, 6)
    d = d.cpu().numpy(); s = s.cpu().numpy()
    assert np.isfinite(d).all() and np.isfinite(s).all()
    assert set(np.unique(d[:, :, 19]).astype(int)).issubset({0, 4, 5})
    nd = d[:, :, 20].astype(int)  # doublings computed; each computes 2**i steps unless cut by a sub-U-turn
    steps = d[:, :, 6]
    full = d[:, :, 19] != 5
    assert (steps[full] == (2 ** nd[full] - 1)).all()
    assert (steps[~full] <= (2 ** nd[~full] - 1)).all() and (steps[~full] > 2 ** (nd[~full] - 1) - 1).all()
    assert (d[:, :, 7] == 0).all()
    assert d[:, :, 17].max() < 5.0  # orbit energy error stays O(1) at H0=0.5 (~1.25 max)
    assert (d[:, :, 4] - d[:, :, 5] == steps).all()  # maxFint - maxBint = number of steps
    # draws: second generated component is sum(q^2) of the first... check against recomputation
    s2, d2 = e.run(q0, SEED, 6)
    np.testing.assert_array_equal(s2.cpu().numpy(), s)  # deterministic
    e.close()


def test_error_paths(wa):
    with pytest.raises(ValueError):
        wa.Engine(3, 4, M=0)
    with pytest.raises(ValueError):
        wa.Engine(3, 4, warmupIter=10, adaptHtarget=1.5)  # the reference: sys.exit("bad adaptHtarget")
    e = wa.Engine(5, 4)
    with pytest.raises(ValueError):
        e.set_target(ol.T_FUNNEL10)
    with pytest.raises(wa.WalnutsError):
        e.run(np.zeros((4, 5)), 0, 1)  # no target yet
    e.close()
    with pytest.raises(NotImplementedError):
        wa.Engine(3000, 4).set_target(ol.T_DENSEGAUSS, np.eye(3000).ravel())  # dense Gaussians stop at D = 1024


def test_kernel_names_are_per_engine_and_failed_set_target_leaves_no_target(wa):
    """wn_kernel_name is engine-owned storage (it used to be a shared 8-entry ring), and a wn_set_target that
    fails after the previous target was released leaves the engine without a target instead of the old kernel."""
    first = wa.Engine(100, 16).set_target(ol.T_DENSEGAUSS, np.eye(100).ravel())
    name0 = first.kernel_name
    assert "wn_fast4_kernel" in name0 and "dense" in name0
    others = [wa.Engine(11, 4, integrator=ol.I_ADAPT_R2P).set_target(ol.T_FUNNEL10) for _ in range(12)]
    assert first.lib.wn_kernel_name(first.h).decode() == name0
    assert all("funnel10" in o.kernel_name for o in others)
    for o in others:
        o.close()
    with pytest.raises(ValueError):
        first.set_target(ol.T_FUNNEL10)  # argument error: rejected before anything is touched, target kept
    assert first.lib.wn_kernel_name(first.h).decode() == name0
    s, d = first.run(np.zeros((16, 100)), 1, 1)
    assert s.shape == (16, 2, 100)
    first.close()
    with pytest.raises(ValueError):
        wa.Engine(3, 4, chain_offset=2 ** 32 - 2)  # global chain ids are 32-bit
    wa.Engine(3, 4, chain_offset=2 ** 32 - 4).close()


def test_host_generated_callable_with_return_torch(wa):
    """A callable `generated` is applied to the stored draws whether numpy arrays or device tensors are returned."""
    from walnuts_amd import adaptiveIntegrators as ai, targetDistr as td
    gen = lambda q: np.array([q[0], np.sum(q * q), q[-1]])
    q0 = np.random.default_rng(2).normal(size=(3, 6))
    kw = dict(generated=gen, integrator=ai.fixedLeapFrog, H0=0.4, numIter=4, warmupIter=0, M=4, seed=3)
    s, d = wa.WALNUTS(td.stdGauss, q0, **kw)
    st, dt = wa.WALNUTS(td.stdGauss, q0, return_torch=True, **kw)
    assert tuple(st.shape) == (3, 3, 5) and st.is_cuda
    np.testing.assert_array_equal(st.cpu().numpy(), s)
    np.testing.assert_array_equal(dt.cpu().numpy(), d)


def test_walnuts_api_shapes(wa):
    """Single-chain call returns exactly the reference's shapes (WALNUTS.py:163,180,724-727)."""
    from walnuts_amd import adaptiveIntegrators as ai, targetDistr as td
    s, d = wa.WALNUTS(td.stdGauss, np.zeros(7), integrator=ai.adaptLeapFrogR2P, numIter=12, warmupIter=0, M=6,
                      H0=0.8, delta0=0.3, seed=5)
    assert s.shape == (7, 13) and d.shape == (12, 24)
    np.testing.assert_array_equal(s[:, 0], 0.0)
    s2, d2 = wa.WALNUTS(td.stdGauss, np.zeros((3, 7)), generated=lambda q: np.array([q[0], sum(q * q)]),
                        numIter=5, warmupIter=0, M=6, H0=0.8, seed=5)
    assert s2.shape == (3, 2, 6) and d2.shape == (3, 5, 24)
    lp, g = td.funnel10(np.arange(11) * 0.1)
    lo, go = ol.lpfun(ol.T_FUNNEL10, np.arange(11) * 0.1)
    np.testing.assert_allclose(lp, lo, rtol=1e-12); np.testing.assert_allclose(g, go, rtol=1e-12)
    with pytest.raises(TypeError):
        wa.WALNUTS("stdGauss", np.zeros(3), warmupIter=0)  # neither a descriptor nor a callable
    with pytest.raises(TypeError):
        wa.WALNUTS(td.stdGauss, np.zeros(3), integrator=lambda *a: None, warmupIter=0)  # integrators are descriptors
    # a plain callable lpFun(q) -> [lp, grad] is the reference's plugin contract: it runs (split mode, host-evaluated)
    s4, d4 = wa.WALNUTS(lambda q: [-0.5 * float(q @ q), -q], np.zeros(3), warmupIter=0, numIter=4, M=4, H0=0.8, seed=2)
    s5, d5 = wa.WALNUTS(td.stdGauss, np.zeros(3), warmupIter=0, numIter=4, M=4, H0=0.8, seed=2)
    close(s4, s5, what="host callable vs fused stdGauss")
    with pytest.raises(SystemExit):
        wa.WALNUTS(td.stdGauss, np.zeros(3), adaptHtarget=2.0)  # WALNUTS.py:140
    # the reference's defaults (warmupIter=1000, adaptH, adaptDelta; WALNUTS.py:119-127) run as they are
    s3, d3 = wa.WALNUTS(td.corrGauss, np.array([1.0, 0.0]), integrator=ai.adaptLeapFrogR2P, numIter=1200, seed=3)
    assert s3.shape == (2, 1201) and d3.shape == (1200, 24)
    assert (d3[1000:, 15] == d3[1000, 15]).all() and (d3[1000:, 18] == d3[1000, 18]).all()  # frozen after warm-up
    assert d3[0, 15] == 0.2 and d3[999, 15] != 0.2 and abs(s3[:, 1000:].mean()) < 0.3


WD = np.load(os.path.join(G, "walnuts_draws.npz"))


@pytest.mark.parametrize("seed", [123, 124, 125, 126])
def test_walnuts_config1_draws_match_reference(wa, seed):
    """BASELINE configs[0]: 10-D standard normal, 1000 draws via walnuts() with test/test.py's
    parameters; the caller's Generator is consumed exactly as the reference does, so the draws
    (and the stream position afterwards) reproduce the reference run."""
    from walnuts_amd import targets as tt
    rng = np.random.default_rng(seed)
    d = wa.walnuts(rng, np.zeros(10), tt.standard_normal_lpdf, tt.standard_normal_grad, np.ones(10),
                   2.0, 10, 0.1, 0, 1000)
    assert d.shape == (1000, 10)
    close(d, WD[f"std10_seed{seed}"], rtol=1e-6, atol=1e-9, what="draws")
    assert rng.uniform() == WD[f"std10_seed{seed}_next_uniform"][0]


def test_walnuts_other_targets_match_reference(wa):
    from walnuts_amd import targets as tt
    rng = np.random.default_rng(7)
    d = wa.walnuts(rng, np.array([0.5, 0.1, -0.2]), tt.funnel_lpdf, tt.funnel_grad, np.array([1.0, 2.0, 0.5]),
                   1.5, 8, 0.2, 20, 150)
    close(d, WD["funnel3_seed7"], what="funnel"); assert rng.uniform() == WD["funnel3_seed7_next_uniform"][0]
    rng = np.random.default_rng(8)
    d = wa.walnuts(rng, np.array([1.0, 0.0]), tt.correlated_normal_lpdf, tt.correlated_normal_grad, np.ones(2),
                   1.8, 8, 0.1, 0, 200)
    close(d, WD["corr2_seed8"], what="corr"); assert rng.uniform() == WD["corr2_seed8_next_uniform"][0]
    rng = np.random.default_rng(9)
    d = wa.walnuts(rng, np.array([0.3, -0.2]), tt.rosenbrock_lpdf, tt.rosenbrock_grad, np.ones(2), 1.0, 8, 0.1, 0, 200)
    close(d, WD["rosen2_seed9"], what="rosenbrock"); assert rng.uniform() == WD["rosen2_seed9_next_uniform"][0]


def test_bench_two_ranks_sharing_one_gpu():
    """bench.py's N > 1 path (self-launch of one child per rank, sharding by chain_offset, MAX-time / SUM-steps
    reductions, final gather) with two ranks on the one GPU of this box; collectives over gloo because RCCL
    refuses two ranks on one device.  `python bench.py --gpus 2` with no launcher around it, and the driver's
    torch.distributed.run form.  The 8-GPU RCCL run is the driver's."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WN_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    bench = os.path.join(root, "bench.py")

    def line(r):
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])

    # self-launch: bench.py starts its own ranks (and the N = 1 reference run of weak_scaling_eff)
    out = line(subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0", "--chains", "256",
                               "--iters", "5", "--n1-ref"], capture_output=True, text=True, timeout=900, env=env, cwd=root))
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["ess_per_sec"]["draws"] == 2 * 256 * 5
    assert len(out["per_rank_kernel_ms"]) == 2 and out["gather_ms"] > 0
    assert out["n1_value_same_invocation"] > 0 and out["weak_scaling_eff"] > 0
    # the driver's form: torch.distributed.run around bench.py gives the same job
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29741", bench, "--gpus", "2", "--steps", "1",
           "--warmup", "0", "--chains", "256", "--iters", "5"]
    out2 = line(subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(env, MASTER_ADDR="127.0.0.1"), cwd=root))
    assert out2["n_gpus"] == 2 and out2["leapfrog_steps_per_launch_per_gpu"] == out["leapfrog_steps_per_launch_per_gpu"]
    # the same 512 chains in one process give the same number of leapfrog steps (sharding invariance)
    o1 = line(subprocess.run([sys.executable, bench, "--steps", "1", "--warmup", "0", "--chains", "512", "--iters", "5",
                              "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=600, env=env, cwd=root))
    assert o1["n_gpus"] == 1 and o1["leapfrog_steps_per_launch_per_gpu"] == 2 * out["leapfrog_steps_per_launch_per_gpu"]
    # --gather identity: the final gather moves the identity draws (SURVEY 8e's message, 8 * C/G * (iters+1) * D bytes per GPU);
    # what rank 0 holds afterwards is bit for bit the draws of the same 512 chains run by one rank
    g2 = line(subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0", "--chains", "256", "--iters", "5",
                              "--gather", "identity", "--no-n1-ref"], capture_output=True, text=True, timeout=600, env=env, cwd=root))
    g1 = line(subprocess.run([sys.executable, bench, "--steps", "1", "--warmup", "0", "--chains", "512", "--iters", "5", "--gather", "identity",
                              "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=600, env=env, cwd=root))
    assert g2["gather_bytes_per_rank"] == 256 * 6 * 100 * 8 and g2["gather_ms"] > 0
    assert g2["draws_sha256"] == g1["draws_sha256"]
    # the other workloads shard the same way (configs 3, 4, 5)
    for extra in (["--workload", "funnel", "--chains", "64", "--iters", "3"],
                  ["--workload", "dense1000", "--chains", "64", "--iters", "2", "--no-cpu-baseline"],
                  ["--workload", "stockwatson", "--chains", "8", "--iters", "2"]):
        o = line(subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-n1-ref"] + extra,
                                capture_output=True, text=True, timeout=600, env=env, cwd=root))
        assert o["n_gpus"] == 2 and o["value"] > 0 and o["roofline"]["frac"] > 0, o
        assert len(o["per_rank_kernel_ms"]) == 2


def test_bench_one_rank_rccl_forced_collectives():
    """RCCL on the one GPU of this box: `--force-collectives` initialises the process group (ncclCommInitRank) with ONE rank
    and runs every collective of the N-rank path on device tensors -- barriers, the MAX / SUM all_reduce, the all_gather of
    the per-rank fields and the final gather of the identity draws (SURVEY 8e's message: 4096 x 31 x 100 doubles = 102 MB).
    Started as a fresh child under torch.distributed.run, exactly as the driver starts ranks.  What rank 0 holds after the
    gather is bit for bit what the plain run produces."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "WN_DIST_BACKEND"):
        env.pop(k, None)
    bench = os.path.join(root, "bench.py")
    common = ["--gpus", "1", "--steps", "1", "--warmup", "0", "--chains", "4096", "--iters", "30", "--gather", "identity",
              "--no-extras", "--no-cpu-baseline"]

    def line(r):
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])

    f = line(subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                             "--master-port", "29751", bench] + common + ["--force-collectives"],
                            capture_output=True, text=True, timeout=900, env=env, cwd=root))
    assert f["n_gpus"] == 1 and f["collectives"].startswith("nccl") and f["gather_ms"] > 0
    assert f["gather_bytes_per_rank"] == 4096 * 31 * 100 * 8 and len(f["per_rank_kernel_ms"]) == 1
    p_ = line(subprocess.run([sys.executable, bench] + common, capture_output=True, text=True, timeout=900, env=env, cwd=root))
    assert p_["collectives"] is None and p_["gather_ms"] == 0
    assert f["draws_sha256"] == p_["draws_sha256"]
    assert f["leapfrog_steps_per_launch_per_gpu"] == p_["leapfrog_steps_per_launch_per_gpu"]
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "bench_1rank_rccl.json"), "w") as fh:
        json.dump(f, fh)


def test_walnuts_batch_config1_four_chains_in_one_call(wa):
    """BASELINE configs[0] in ONE call: the four chains (seeds 123..126) advance together on the device; each reproduces
    the reference's draws and leaves its Generator at the reference's stream position."""
    from walnuts_amd import targets as tt
    rngs = [np.random.default_rng(s) for s in (123, 124, 125, 126)]
    d = wa.walnuts_batch(rngs, np.zeros((4, 10)), tt.standard_normal_lpdf, tt.standard_normal_grad, np.ones(10), 2.0, 10, 0.1, 0, 1000)
    assert d.shape == (4, 1000, 10)
    for k, s in enumerate((123, 124, 125, 126)):
        close(d[k], WD[f"std10_seed{s}"], rtol=1e-9, atol=1e-11, what=f"seed {s}")
        assert rngs[k].uniform() == WD[f"std10_seed{s}_next_uniform"][0]


def test_walnuts_plain_python_callables_as_in_test_py(wa):
    """The reference's (logp, grad) contract (walnuts.py:379-380, test/test.py:11-18): plain Python functions on numpy
    vectors.  Same Generator consumption and draws as the device descriptor of the same target, and as the reference."""
    def lpdf(x):
        return -0.5 * np.dot(x, x)

    def grad(x):
        return -x

    rng = np.random.default_rng(123)
    d = wa.walnuts(rng, np.zeros(10), lpdf, grad, np.ones(10), 2.0, 10, 0.1, 0, 25)
    close(d, WD["std10_seed123"][:25], rtol=1e-9, atol=1e-11, what="plain callables")
    # a target with no device form at all, any dimension
    A = np.diag(np.linspace(1.0, 4.0, 70))
    rng1, rng2 = np.random.default_rng(5), np.random.default_rng(5)
    d1 = wa.walnuts(rng1, np.ones(70), lambda x: -0.5 * x @ A @ x, lambda x: -A @ x, np.ones(70), 0.9, 6, 0.3, 2, 6)
    from walnuts_amd import targetDistr as td
    d2 = wa.walnuts(rng2, np.ones(70), *(2 * [td.diagGauss(np.diag(A))]), np.ones(70), 0.9, 6, 0.3, 2, 6)
    close(d1, d2, rtol=1e-9, atol=1e-11, what="callable vs device target, D = 70")
    assert rng1.uniform() == rng2.uniform()


def test_walnuts_stan_shape(wa):
    with pytest.raises(ImportError, match="bridgestan"):
        wa.walnuts_stan(np.random.default_rng(0), np.zeros(3), "m.stan", "d.json", np.ones(3), 1.0, 5, 0.1, 0, 10)


def test_walnuts_batch_evaluates_only_the_chains_that_asked(wa):
    """Host (logp, grad) callables in a batch: a chain whose extension has finished keeps a stale row in the request buffer;
    only the chains that asked in a round are evaluated (wn_wstep_ext_waiting), so four chains in one call make exactly the
    calls of the four chains run one by one -- the points the reference evaluates (walnuts.py:144-276)."""
    calls = {"n": 0}

    def lpdf(x):
        calls["n"] += 1
        assert np.all(np.isfinite(x))
        return -0.5 * np.dot(x, x)

    def grad(x):
        return -x

    th = np.random.default_rng(9).normal(size=(4, 6))
    single = 0
    outs = []
    for c in range(4):
        calls["n"] = 0
        outs.append(wa.walnuts(np.random.default_rng(100 + c), th[c], lpdf, grad, np.ones(6), 1.3, 6, 0.2, 0, 5))
        single += calls["n"]
    calls["n"] = 0
    d = wa.walnuts_batch([np.random.default_rng(100 + c) for c in range(4)], th, lpdf, grad, np.ones(6), 1.3, 6, 0.2, 0, 5)
    assert calls["n"] == single, (calls["n"], single)
    for c in range(4):
        close(d[c], outs[c], rtol=1e-12, atol=0, what=f"chain {c}")


def test_walnuts_device_whole_transitions_on_the_device(wa):
    """walnuts_device: walnuts_step's transition kernel (walnuts.py:279-359) with the engine's Philox stream in place of numpy
    Generators.  Reproducible, independent of batching / sharding and of the launch granularity, and it samples the target:
    512 chains x 60 transitions of the 10-D standard normal with test.py's parameters."""
    from walnuts_amd import targetDistr as td
    th0 = np.zeros((512, 10))
    args = (td.stdGauss, np.ones(10), 2.0, 10, 0.1, 5, 60)
    d = wa.walnuts_device(77, th0, *args)
    assert d.shape == (512, 60, 10) and np.isfinite(d).all()
    np.testing.assert_array_equal(d, wa.walnuts_device(77, th0, *args, transitions_per_launch=7))
    lo = wa.walnuts_device(77, th0[:200], *args); hi = wa.walnuts_device(77, th0[200:], *args, chain_offset=200)
    np.testing.assert_array_equal(d, np.concatenate([lo, hi]))
    x = d[:, 10:, :]
    assert np.abs(x.mean(axis=(0, 1))).max() < 0.03, x.mean(axis=(0, 1))
    assert np.abs(x.var(axis=(0, 1)) - 1.0).max() < 0.05, x.var(axis=(0, 1))
    # a different seed is a different stream
    assert not np.array_equal(d, wa.walnuts_device(78, th0, *args))
    # the same kernel as the host-Generator form: same one-transition law (first two moments of the first transition)
    rngs = [np.random.default_rng(c) for c in range(512)]
    h = wa.walnuts_batch(rngs, th0, td.stdGauss, td.stdGauss, np.ones(10), 2.0, 10, 0.1, 0, 1)[:, 0]
    one = wa.walnuts_device(5, th0, td.stdGauss, np.ones(10), 2.0, 10, 0.1, 0, 1)[:, 0]
    assert abs(h.var() - one.var()) < 0.15 * h.var(), (h.var(), one.var())


def test_walnuts_device_wave_kernel_matches_the_one_lane_kernel(wa):
    """Round 4: whole transitions of Gaussian targets run with the wave doing the work (lane = element, wstep_wave_kernel); the one-lane
    state machine (wstep_auto_kernel, route WN_ROUTE_WSTEP_LEGACY) is the same transition kernel (walnuts.py:279-359) on the same random stream.
    The two associate their sums differently, so a chain agrees to rounding until a comparison at the edge of a threshold goes the
    other way (stable_steps' `<= max_error`, the acceptance test): nearly every chain must agree, and the moments must."""
    from walnuts_amd import targetDistr as td
    for D, tgt, im, C, n in ((10, td.stdGauss, np.ones(10), 384, 16), (24, td.diagGauss(1.0 / np.logspace(0, 1, 24)), np.logspace(0, 1, 24), 128, 8)):
        th0 = np.random.default_rng(3).normal(size=(C, D))
        args = (tgt, im, 2.0, 8, 0.1, 0, n)
        new = wa.walnuts_device(91, th0, *args)
        old = wa.walnuts_device(91, th0, *args, route="WSTEP_LEGACY")
        assert np.isfinite(new).all()
        same = np.abs(new - old).max(axis=(1, 2)) < 1e-8
        assert same.mean() > 0.95, same.mean()
        first = np.abs(new[:, 0] - old[:, 0]).max(axis=1) < 1e-9  # the first transition alone: hardly any threshold is that close
        assert first.mean() > 0.99, first.mean()


@pytest.mark.parametrize("legacy", [False, True], ids=["wave_kernel", "one_lane_kernel"])
def test_walnuts_device_matches_the_reference_on_the_engine_stream(wa, legacy):
    """Pins `walnuts_device` (wn_wstep_run_device_rng) to the reference: tests/golden/walnuts_device.npz holds the UNMODIFIED
    /root/reference/walnuts/walnuts.py::walnuts() driven by a Generator stand-in that serves the engine's Philox stream in the
    kernels' draw order (tests/golden/make_golden_walnuts_device.py), 128 chains x 30 transitions of the 10-D standard normal with
    test/test.py's parameters, a diagonal-mass case and a 24-D diagonal-precision Gaussian.  Both device kernels -- the wave kernel
    (wstep_wave_kernel) and the one-lane state machine (wstep_auto_kernel) -- must reproduce every transition: doublings started,
    accepts, the sum of the chosen micro-step counts and of the forward ell_stable EXACTLY, every draw to rtol 1e-6.  No chain is
    budgeted away: a chain may only leave through a `<= max_error` / acceptance comparison sitting on a rounding edge, and there is
    none in these 4 672 transitions (128 x 30 + 32 x 20 + 16 x 12)."""
    from walnuts_amd import targetDistr as td
    g = np.load(os.path.join(G, "walnuts_device.npz"))
    if True:
        for name in ("std10", "std10_mass", "diag24"):
            seed, C, T, D, depth, it0 = (int(v) for v in g[name + "_cfg"])
            macro, maxerr = (float(v) for v in g[name + "_par"])
            prec = g[name + "_prec"]
            tgt = td.stdGauss if np.all(prec == 1.0) else td.diagGauss(prec)
            d, st = wa.walnuts_device(seed, g[name + "_theta0"], tgt, g[name + "_inv_mass"], macro, depth, maxerr, 0, T, iter0=it0,
                                      return_stats=True, transitions_per_launch=11, route="WSTEP_LEGACY" if legacy else 0)
            ref_i = g[name + "_ints"][:, :, :4]
            bad_i = np.flatnonzero((st != ref_i).any(axis=(1, 2)))
            assert bad_i.size == 0, (name, "chains whose integers differ from the reference", bad_i[:10], st[bad_i[:1]], ref_i[bad_i[:1]])
            ref_d = g[name + "_draws"]
            err = np.abs(d - ref_d) / (1e-9 + 1e-6 * np.abs(ref_d))
            bad_d = np.flatnonzero((err > 1.0).any(axis=(1, 2)))
            assert bad_d.size == 0, (name, "chains whose draws leave rtol 1e-6", bad_d[:10], err.max())


def test_walnuts_batch_wave_kernels_match_the_thread_per_chain_kernels(wa):
    """Round 5: walnuts_batch() -- the reference's contract, one numpy Generator per chain consumed on the host -- runs its extensions
    on the wave kernels (wstep_wv_*: lane = element, the chain's state in memory between the launches of a doubling) for the
    element-wise Gaussian targets.  Same Generators, same draws as the thread-per-chain kernels (route WN_ROUTE_WSTEP_LEGACY), which the
    reference goldens pin for every target: D = 24 > 16 (full-wave sums), a diagonal mass matrix, ragged activity (chains stop at
    different depths), and every Generator ends at the same stream position."""
    from walnuts_amd import targetDistr as td
    D, C, T = 24, 96, 6
    prec = np.geomspace(0.5, 8.0, D)
    im = np.geomspace(0.5, 2.0, D)
    th0 = np.random.default_rng(4).normal(size=(C, D)) / np.sqrt(prec)
    tgt = td.diagGauss(prec)
    args = (tgt, tgt, im, 1.5, 7, 0.2, 0, T)
    ra = [np.random.default_rng(900 + c) for c in range(C)]
    rb = [np.random.default_rng(900 + c) for c in range(C)]
    new = wa.walnuts_batch(ra, th0, *args)
    old = wa.walnuts_batch(rb, th0, *args, route="WSTEP_LEGACY")
    assert np.isfinite(new).all()
    close(new, old, rtol=1e-9, atol=1e-12, what="wave host-protocol kernels vs thread-per-chain kernels")
    assert all(ra[c].uniform() == rb[c].uniform() for c in range(C))
