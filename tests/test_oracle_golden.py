"""Pin the CPU oracle against outputs of the unmodified Python reference.

The fixtures in tests/golden/*.npz were produced by tests/golden/make_golden.py running
/root/reference/WALNUTSpy with a tape of the engine's RNG stream (see that script).  These
tests replay the same tapes through oracle/libwalnuts_oracle.so and require agreement on every
integrator call of every iteration: integers exactly, reals to 1e-12 (libm vs NumPy exp/log
may differ in the last bit).
"""
import os

import numpy as np
import pytest

import oracle_lib as ol

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TR = np.load(os.path.join(G, "walnuts_traces.npz"))
KA = np.load(os.path.join(G, "integrator_kats.npz"))
CASES = [str(x) for x in TR["case_names"]]
RT = 1e-12
EXP_TARGETS = (ol.T_FUNNEL10, ol.T_FUNNEL1, ol.T_MODFUNNEL)


def close(a, b, rtol=RT, atol=1e-13):
    a = np.asarray(a, float); b = np.asarray(b, float)
    assert a.shape == b.shape, (a.shape, b.shape)
    both_nan = np.isnan(a) & np.isnan(b)
    same_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))
    ok = both_nan | same_inf | (np.abs(a - b) <= atol + rtol * np.abs(b))
    assert ok.all(), f"max abs diff {np.nanmax(np.abs(a - b))} at {np.argwhere(~ok)[:5]}"


def load_case(name):
    p = name + "__"
    cfg = TR[p + "cfg"]; par = TR[p + "par"]
    c = dict(target=int(cfg[0]), integrator=int(cfg[1]), M=int(cfg[2]), numIter=int(cfg[3]),
             minC=int(cfg[4]), maxC=int(cfg[5]), generated=int(cfg[6]), gen_k=int(cfg[7]),
             chain=int(cfg[8]), H0=float(par[0]), stepSizeRandScale=float(par[1]),
             delta0=float(par[2]), R2Pprob0=float(par[3]), q0=TR[p + "q0"],
             tpar=TR[p + "tpar"] if (p + "tpar") in TR.files else None)
    w = TR[p + "warm"] if (p + "warm") in TR.files else np.array([0, 1, 0.8, 1, 0.6, 0.9])
    c.update(warmupIter=int(w[0]), adaptH=bool(w[1]), adaptHtarget=float(w[2]), adaptDelta=bool(w[3]),
             adaptDeltaTarget=float(w[4]), adaptDeltaQuantile=float(w[5]))
    ref = {k: TR[p + k] for k in ("samples", "diagnostics", "trace", "tape_u", "tape_z", "n_u")}
    if (p + "orbitMin") in TR.files:
        ref["orbitMin"] = TR[p + "orbitMin"]; ref["orbitMax"] = TR[p + "orbitMax"]
    return c, ref


def run_oracle(c, ref, tape=True):
    kw = dict(integrator=c["integrator"], tpar=c["tpar"], H0=c["H0"],
              stepSizeRandScale=c["stepSizeRandScale"], delta0=c["delta0"], numIter=c["numIter"],
              M=c["M"], minC=c["minC"], maxC=c["maxC"], R2Pprob0=c["R2Pprob0"],
              generated=c["generated"], gen_k=c["gen_k"], trace_cap=len(ref["trace"]) + 8,
              warmupIter=c["warmupIter"], adaptH=c["adaptH"], adaptHtarget=c["adaptHtarget"],
              adaptDelta=c["adaptDelta"], adaptDeltaTarget=c["adaptDeltaTarget"],
              adaptDeltaQuantile=c["adaptDeltaQuantile"], recordOrbitStats="orbitMin" in ref)
    if tape:
        tu = np.nan_to_num(ref["tape_u"], nan=0.5)
        kw.update(tape_u=tu, tape_z=ref["tape_z"])
    else:
        kw.update(seed=int(TR["seed"][0]), chain=c["chain"])
    return ol.walnuts(c["target"], c["q0"], **kw)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("tape", [True, False], ids=["tape", "philox"])
def test_walnuts_trace_matches_reference(name, tape):
    c, ref = load_case(name)
    out = run_oracle(c, ref, tape)
    # same number of integrator calls and of uniforms consumed per iteration
    assert out["ncalls"] == len(ref["trace"])
    np.testing.assert_array_equal(out["n_u"], ref["n_u"])
    tr, rt = out["trace"], ref["trace"]
    # integer fields: iter, xi, If, Ib, c, nEvalF, nEvalB (col 1 = doubling index, oracle-only)
    for col in (0, 2, 6, 7, 8, 10, 11):
        np.testing.assert_array_equal(tr[:, col], rt[:, col], err_msg=f"trace col {col}")
    for col in (3, 4, 5, 9):
        close(tr[:, col], rt[:, col], rtol=1e-9, atol=1e-11)
    if c["target"] in EXP_TARGETS or c["warmupIter"] > 0:
        # NumPy's SIMD exp/log and glibc's differ in the last bit; the funnel dynamics amplify it
        # (with warm-up the adapted H = delta**(1/3)*exp(P2 quantile of log igrConst) carries the same
        # last-bit difference into every target)
        close(tr[:, 12:], rt[:, 12:], rtol=1e-9, atol=1e-11)
        close(out["samples"], ref["samples"], rtol=1e-9, atol=1e-11)
    else:
        # no transcendental in the dynamics: the restatement is bit-exact
        np.testing.assert_array_equal(tr[:, 12:], rt[:, 12:])  # per-step (q, v)
        np.testing.assert_array_equal(out["samples"], ref["samples"])
        np.testing.assert_array_equal(tr[:, 3:6], rt[:, 3:6])
    if "orbitMin" in ref:
        close(out["orbitMin"], ref["orbitMin"], rtol=1e-9, atol=1e-11)
        close(out["orbitMax"], ref["orbitMax"], rtol=1e-9, atol=1e-11)
    d, rd = out["diagnostics"], ref["diagnostics"]
    for col in (0, 1, 4, 5, 6, 7, 8, 9, 12, 13, 19, 20, 21, 22):
        np.testing.assert_array_equal(d[:, col], rd[:, col], err_msg=f"diag col {col}")
    close(d, rd, rtol=1e-9, atol=1e-11)


def test_case_coverage():
    """The fixture set exercises every stop code and both quirk paths."""
    codes = set()
    for n in CASES:
        codes |= set(TR[n + "__diagnostics"][:, 19].astype(int).tolist())
    assert {4, 5, -4, 0, 999}.issubset(codes)
    assert any(np.isnan(TR[n + "__diagnostics"][:, 17]).any() or
               np.isinf(TR[n + "__diagnostics"][:, 17]).any() for n in CASES)  # forcedReject


@pytest.mark.parametrize("i", range(int(KA["n"][0])))
def test_integrator_kat(i):
    p = f"k{i}__"
    tk, kind, d, xi, minC, maxC = [int(x) for x in KA[p + "meta"]]
    Ham0, h, delta, u = [float(x) for x in KA[p + "par"]]
    tpar = KA["dense6_A"] if tk == ol.T_DENSEGAUSS else None
    o = ol.integrator(kind, tk, KA[p + "q"], KA[p + "v"], KA[p + "g"], Ham0, h, xi, delta,
                      tpar=tpar, minC=minC, maxC=maxC, u_r2p=u)
    np.testing.assert_array_equal([o["nEvalF"], o["nEvalB"], o["If"], o["Ib"], o["c"]], KA[p + "io"])
    close([o["lp"], o["lwt"], o["igrConst"]], KA[p + "do"], rtol=1e-11)
    close(o["q"], KA[p + "q1"]); close(o["v"], KA[p + "v1"]); close(o["grad"], KA[p + "g1"])


@pytest.mark.parametrize("tname", ["std3", "std100", "funnel", "dense6", "corr", "smile", "funnel1",
                                   "modfunnel"])
def test_lpfun(tname):
    tk, d = [int(x) for x in KA[f"lp__{tname}__kind"]]
    tpar = KA["dense6_A"] if tk == ol.T_DENSEGAUSS else None
    for q, lp, g in zip(KA[f"lp__{tname}__q"], KA[f"lp__{tname}__lp"], KA[f"lp__{tname}__g"]):
        l2, g2 = ol.lpfun(tk, q, tpar)
        close(l2, lp, rtol=1e-13); close(g2, g, rtol=1e-13)


@pytest.mark.parametrize("k", range(3))
def test_p2quantile_matches_reference(k):
    """P2quantile.py:16-86, bit for bit on every push."""
    got = ol.p2quantile(float(KA[f"p2__{k}__prob"][0]), KA[f"p2__{k}__x"])
    np.testing.assert_array_equal(got, KA[f"p2__{k}__q"])


def test_np_quantile_matches_numpy():
    for i in range(int(KA["quant__n"][0])):
        q, r = KA[f"quant__{i}__qr"]
        assert ol.np_quantile(KA[f"quant__{i}__a"], float(q)) == r


def test_stdgauss_lp_bitexact_pairwise_sum():
    """np.sum's pairwise order (targetDistr.py:19) is reproduced bit for bit."""
    rng = np.random.default_rng(5)
    for n in list(range(1, 40)) + [100, 127, 128, 129, 200, 300, 1000]:
        for _ in range(10):
            q = rng.normal(size=n) * 10 ** rng.uniform(-2, 2, size=n)
            lp, _ = ol.lpfun(ol.T_STDGAUSS, q)
            assert lp == -0.5 * np.sum(q * q)


@pytest.mark.parametrize("i", range(7))
def test_subtreeplan(i):
    want = KA[f"plan__{i}"]
    got = ol.subtreeplan(2 ** i)
    if 2 ** i == 1:
        assert len(got) == 0
    else:
        np.testing.assert_array_equal(got, want)


def test_stopcondition():
    for i in range(int(KA["stop__n"][0])):
        a = KA[f"stop__{i}__args"]
        assert ol.stopcondition(*a) == bool(KA[f"stop__{i}__res"][0])


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    assert ol.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert ol.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6,
                                                            0x6d5451fd]
    assert ol.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
