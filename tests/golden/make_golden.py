#!/usr/bin/env python3
"""Generate golden fixtures by running the UNMODIFIED Python reference in this container.

Run:  PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python3 -B tests/golden/make_golden.py

The reference (/root/reference/WALNUTSpy) holds no golden vectors of its own
(SURVEY.md section 4), and it draws from NumPy's global legacy stream in a data-dependent
order (quirk Q1).  To make the reference, the C oracle and the GPU engine consume identical
randomness, numpy.random.uniform / numpy.random.normal are replaced -- for the duration of
one reference call -- by a *tape* that serves the engine's own counter-based stream
(Philox4x32-10 keyed by (seed; chain, iteration, stream), DESIGN.md "RNG") in call order.
Nothing of the reference is copied: the fixtures hold only inputs (configuration, q0,
tapes) and the reference's outputs (samples, diagnostics, per-integrator-call trace).

The fixtures are data; this script is how they were made.  It cannot run on the GPU box
(/root/reference does not exist there) and no test imports it.
"""
import importlib.util
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402  (only for the engine RNG tape + enums)

REF = "/root/reference/WALNUTSpy"
sys.path.insert(0, REF)
import WALNUTS as wn  # noqa: E402
import adaptiveIntegrators as ai  # noqa: E402
import targetDistr as td  # noqa: E402

SEED = 20251121  # SURVEY.md 8d master seed


class Tape:
    """Serves the engine stream to the reference through numpy.random.{uniform,normal}."""

    def __init__(self, seed, chain, M, d):
        self.seed, self.chain, self.M, self.d = seed, chain, M, d
        self.iter = 0
        self.k = 0
        self.u_used = []  # per iteration list of uniforms consumed
        self.z_used = []
        self._cache = None

    def _u(self, k):
        if self._cache is None or k >= len(self._cache):
            n = 256 if self._cache is None else 2 * len(self._cache)
            while n <= k:
                n *= 2
            self._cache, _ = ol.rng_tape(self.seed, self.chain, self.iter, n, 0)
        return self._cache[k]

    def uniform(self, low=0.0, high=1.0, size=None):
        # WALNUTS.py:216 is the first draw of every iteration
        if size is not None and size == self.M and low == 0.0 and high == 2.0:
            self.iter += 1
            self.k = 0
            self._cache = None
            self.u_used.append([])
        n = 1 if size is None else int(size)
        us = np.array([self._u(self.k + j) for j in range(n)])
        self.k += n
        self.u_used[-1].extend(us.tolist())
        vals = low + (high - low) * us  # numpy legacy: low + (high-low)*random_double
        return np.float64(vals[0]) if size is None else vals

    def normal(self, size=None):
        assert size == self.d
        _, z = ol.rng_tape(self.seed, self.chain, self.iter, 0, size)
        self.z_used.append(z.copy())
        return z.copy()


def run_reference(lpFun, q0, integrator, seed=SEED, chain=0, generated=None, **kw):
    M = kw.get("M", 10)
    d = q0.size
    tape = Tape(seed, chain, M, d)
    trace = []
    state = {"iter": 0}

    def logged(q, v, g, Ham0, h, xi, lpF, delta, aux):
        out = integrator(q, v, g, Ham0, h, xi, lpF, delta, aux)
        row = [tape.iter, -1, xi, h, Ham0, out.lp, out.If, out.Ib, out.c, out.lwt, out.nEvalF,
               out.nEvalB]
        trace.append(np.concatenate([row, np.asarray(out.q, float), np.asarray(out.v, float)]))
        return out

    saved = (np.random.uniform, np.random.normal)
    np.random.uniform, np.random.normal = tape.uniform, tape.normal
    try:
        args = dict(kw)
        if generated is not None:
            args["generated"] = generated
        ret = wn.WALNUTS(lpFun, q0.copy(), integrator=logged, **args)
        samples, diag = ret[0], ret[1]
    finally:
        np.random.uniform, np.random.normal = saved
    numIter = kw["numIter"]
    nu = np.array([len(u) for u in tape.u_used], dtype=np.int32)
    tu = np.full((numIter, int(nu.max())), np.nan)
    for i, u in enumerate(tape.u_used):
        tu[i, :len(u)] = u
    tz = np.array(tape.z_used)
    res = dict(samples=samples, diagnostics=diag, trace=np.array(trace), tape_u=tu, tape_z=tz, n_u=nu)
    if len(ret) == 4:
        res["orbitMin"], res["orbitMax"] = ret[2], ret[3]
    return res


# ---- targets that are not in the reference, as reference-convention callables
def make_dense(A):
    A = np.asarray(A, float)
    d = A.shape[0]

    def lp(q, hessian=False):
        g = np.empty(d)
        for i in range(d):
            a = 0.0
            for j in range(d):
                a += float(A[i, j]) * float(q[j])
            g[i] = -a
        s = 0.0
        for i in range(d):
            s += float(q[i]) * float(g[i])
        return [0.5 * s, g]
    return lp


def make_diag(a):
    a = np.asarray(a, float)

    def lp(q, hessian=False):
        g = -(a * q)
        s = 0.0
        for i in range(a.size):
            s += float(q[i]) * float(g[i])
        return [0.5 * s, g]
    return lp


def dense_precision(d, kappa, seed):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.normal(size=(d, d)))
    lam = np.logspace(0, np.log10(kappa), d)
    A = (Q * (1.0 / lam)) @ Q.T
    A = 0.5 * (A + A.T)
    Sig_half = Q * np.sqrt(lam)
    return A, Sig_half


INTEGRATORS = {"fixed": (ai.fixedLeapFrog, ol.I_FIXED), "D": (ai.adaptLeapFrogD, ol.I_ADAPT_D),
               "R2P": (ai.adaptLeapFrogR2P, ol.I_ADAPT_R2P), "Y": (ai.adaptYoshidaD, ol.I_ADAPT_YOSHIDA),
               "F": (ai.adaptLeapFrogFlowD, ol.I_ADAPT_FLOWD)}


def gen_q0_sumsq(q):
    return np.array([q[0], sum(q * q)])


def main():
    out = {}
    rng = np.random.default_rng(7)
    cases = []

    def add(name, target_kind, lpFun, q0, integ, tpar=None, generated=ol.G_IDENTITY, gen_k=0,
            genfn=None, minC=0, maxC=10, warm=None, **kw):
        cases.append(dict(name=name, target=target_kind, lpFun=lpFun, q0=np.asarray(q0, float),
                          integ=integ, tpar=tpar, generated=generated, gen_k=gen_k, genfn=genfn,
                          minC=minC, maxC=maxC, warm=warm or {}, kw=kw))

    for ig in ("fixed", "D", "R2P"):
        add(f"std3_{ig}", ol.T_STDGAUSS, td.stdGauss, rng.normal(size=3), ig,
            H0=0.9, delta0=0.05, numIter=30, M=7)
        add(f"funnel_{ig}", ol.T_FUNNEL10, td.funnel10,
            np.r_[1.2, np.exp(0.6) * rng.normal(size=10)], ig, H0=0.3, delta0=0.3, numIter=25, M=8)
        A8, S8 = dense_precision(8, 100.0, 3)
        add(f"dense8_{ig}", ol.T_DENSEGAUSS, make_dense(A8), S8 @ rng.normal(size=8), ig,
            tpar=A8.ravel(), H0=0.5, delta0=0.1, numIter=20, M=8)
    add("std100_fixed", ol.T_STDGAUSS, td.stdGauss, rng.normal(size=100), "fixed",
        H0=1.4 * 100 ** -0.25, numIter=12, M=6, generated=ol.G_Q0_SUMSQ, genfn=gen_q0_sumsq)
    add("std20_R2P", ol.T_STDGAUSS, td.stdGauss, rng.normal(size=20), "R2P",
        H0=1.4 * 20 ** -0.25, delta0=0.3, numIter=20, M=8, generated=ol.G_Q0_SUMSQ,
        genfn=gen_q0_sumsq)
    a5 = 1.0 / np.logspace(0, 3, 5)
    add("diag5_D", ol.T_DIAGGAUSS, make_diag(a5), rng.normal(size=5) / np.sqrt(a5), "D",
        tpar=a5, H0=0.8, delta0=0.2, numIter=20, M=8)
    add("std4_D_minC3", ol.T_STDGAUSS, td.stdGauss, rng.normal(size=4), "D",
        H0=1.0, delta0=0.3, numIter=15, M=6, minC=3)
    add("std4_R2P_maxC2", ol.T_STDGAUSS, td.stdGauss, rng.normal(size=4), "R2P",
        H0=2.5, delta0=0.01, numIter=25, M=6, maxC=2)
    add("funnel_R2P_bigH", ol.T_FUNNEL10, td.funnel10,
        np.r_[-3.0, np.exp(-1.5) * rng.normal(size=10)], "R2P", H0=2.0, delta0=0.5, numIter=20,
        M=6, maxC=4, generated=ol.G_FIRSTK, gen_k=2, genfn=lambda q: np.array([q[0], q[1]]))
    add("funnel_fixed_blowup", ol.T_FUNNEL10, td.funnel10,
        np.r_[-6.0, np.exp(-3.0) * rng.normal(size=10)], "fixed", H0=4.0, numIter=15, M=6)
    add("funnel_fixed_blowup2", ol.T_FUNNEL10, td.funnel10,
        np.r_[-9.0, np.exp(-4.5) * rng.normal(size=10)], "fixed", H0=9.0, numIter=15, M=5)
    add("funnel_D_blowup", ol.T_FUNNEL10, td.funnel10,
        np.r_[-9.0, np.exp(-4.5) * rng.normal(size=10)], "D", H0=9.0, delta0=0.3, numIter=8, M=4,
        maxC=3)
    add("std3_fixed_M3", ol.T_STDGAUSS, td.stdGauss, rng.normal(size=3), "fixed",
        H0=0.1, numIter=20, M=3)
    add("std3_R2P_M2", ol.T_STDGAUSS, td.stdGauss, rng.normal(size=3), "R2P",
        H0=0.2, delta0=0.3, numIter=20, M=2)
    # warm-up adaptation of H and delta (WALNUTS.py:701-712 + P2quantile.py), SURVEY.md 8f rank 1
    add("corr_R2P_warm", ol.T_CORRGAUSS, td.corrGauss, np.array([1.0, 0.0]), "R2P",
        H0=0.2, delta0=0.05, numIter=70, M=7, warm=dict(warmupIter=45))
    add("funnel_R2P_warm", ol.T_FUNNEL10, td.funnel10,
        np.r_[0.7, np.exp(0.35) * rng.normal(size=10)], "R2P", H0=0.3, delta0=0.3, numIter=50, M=8,
        warm=dict(warmupIter=35, adaptDeltaTarget=0.6))
    add("std3_fixed_warm", ol.T_STDGAUSS, td.stdGauss, rng.normal(size=3), "fixed",
        H0=0.3, delta0=0.05, numIter=45, M=7, warm=dict(warmupIter=30, adaptHtarget=0.7))
    A8w, S8w = dense_precision(8, 100.0, 3)
    add("dense8_D_warm", ol.T_DENSEGAUSS, make_dense(A8w), S8w @ rng.normal(size=8), "D",
        tpar=A8w.ravel(), H0=0.4, delta0=0.1, numIter=40, M=7,
        warm=dict(warmupIter=25, adaptDelta=False))
    add("std4_D_warm_noH", ol.T_STDGAUSS, td.stdGauss, rng.normal(size=4), "D",
        H0=0.8, delta0=0.2, numIter=40, M=6, warm=dict(warmupIter=30, adaptH=False, adaptDeltaQuantile=0.8))
    # recordOrbitStats (WALNUTS.py:182-184, 331-333), SURVEY.md 8f rank 3
    add("funnel_R2P_orbit", ol.T_FUNNEL10, td.funnel10, np.r_[-1.0, np.exp(-0.5) * rng.normal(size=10)], "R2P",
        H0=0.3, delta0=0.3, numIter=20, M=8, generated=ol.G_FIRSTK, gen_k=2,
        genfn=lambda q: np.array([q[0], q[1]]), recordOrbitStats=True)
    add("std3_fixed_orbit", ol.T_STDGAUSS, td.stdGauss, rng.normal(size=3), "fixed",
        H0=0.6, numIter=25, M=7, recordOrbitStats=True)
    # adaptYoshidaD (adaptiveIntegrators.py:142-240), SURVEY.md 8f rank 2
    add("std3_Y", ol.T_STDGAUSS, td.stdGauss, rng.normal(size=3), "Y", H0=1.5, delta0=0.01, numIter=25, M=6)
    add("funnel_Y", ol.T_FUNNEL10, td.funnel10, np.r_[0.4, np.exp(0.2) * rng.normal(size=10)], "Y",
        H0=0.5, delta0=0.05, numIter=15, M=6)
    add("dense8_Y_warm", ol.T_DENSEGAUSS, make_dense(A8w), S8w @ rng.normal(size=8), "Y",
        tpar=A8w.ravel(), H0=0.8, delta0=0.05, numIter=30, M=6, warm=dict(warmupIter=20))
    add("corr_R2P", ol.T_CORRGAUSS, td.corrGauss, np.array([1.0, 0.0]), "R2P",
        H0=0.7, delta0=0.1, numIter=25, M=7)
    add("smile_D", ol.T_SMILE, td.smileDistr, np.array([0.3, -0.2]), "D",
        H0=0.5, delta0=0.1, numIter=25, M=7)
    add("funnel1_R2P", ol.T_FUNNEL1, td.funnel1, np.array([0.5, 0.1]), "R2P",
        H0=0.6, delta0=0.2, numIter=25, M=7)
    add("modfunnel_fixed", ol.T_MODFUNNEL, td.modFunnel, np.array([0.2, 0.4]), "fixed",
        H0=0.3, numIter=25, M=7)
    # adaptLeapFrogFlowD (adaptiveIntegrators.py:246-356), SURVEY.md 8f rank 2 (appended: earlier cases keep their draws)
    add("std3_F", ol.T_STDGAUSS, td.stdGauss, rng.normal(size=3), "F", H0=1.2, delta0=0.01, numIter=25, M=6)
    add("funnel_F", ol.T_FUNNEL10, td.funnel10, np.r_[0.3, np.exp(0.15) * rng.normal(size=10)], "F",
        H0=0.4, delta0=0.02, numIter=15, M=6)
    add("smile_F", ol.T_SMILE, td.smileDistr, np.array([0.4, -0.1]), "F", H0=0.6, delta0=0.005, numIter=25, M=7)
    add("dense8_F_warm", ol.T_DENSEGAUSS, make_dense(A8w), S8w @ rng.normal(size=8), "F",
        tpar=A8w.ravel(), H0=0.8, delta0=0.02, numIter=30, M=6, warm=dict(warmupIter=20))
    # recordOrbitStats with generated = [q[0], sum(q*q)] (WALNUTS.py:182-184,331-333 with mainGaussESS.py:29-30), SURVEY.md 8f
    # rank 3 (appended: earlier cases keep their draws)
    add("std6_fixed_orbit_ess", ol.T_STDGAUSS, td.stdGauss, rng.normal(size=6), "fixed",
        H0=0.7, numIter=25, M=7, generated=ol.G_Q0_SUMSQ, genfn=gen_q0_sumsq, recordOrbitStats=True)
    add("dense8_R2P_orbit_ess", ol.T_DENSEGAUSS, make_dense(A8w), S8w @ rng.normal(size=8), "R2P",
        tpar=A8w.ravel(), H0=0.6, delta0=0.1, numIter=20, M=7, generated=ol.G_Q0_SUMSQ, genfn=gen_q0_sumsq, recordOrbitStats=True)

    for chain, c in enumerate(cases):
        fn, kind = INTEGRATORS[c["integ"]]
        aux = ai.integratorAuxPar(minC=c["minC"], maxC=c["maxC"])
        kw = dict(c["kw"])
        kw.setdefault("delta0", 0.05)
        wk = dict(warmupIter=0, adaptH=True, adaptHtarget=0.8, adaptDelta=True, adaptDeltaTarget=0.6,
                  adaptDeltaQuantile=0.9)  # WALNUTS.py:120-127 defaults (warm-up off unless asked)
        wk.update(c["warm"])
        res = run_reference(c["lpFun"], c["q0"], fn, chain=chain, generated=c["genfn"],
                            igrAux=aux, **wk, **kw)
        out[c["name"] + "__warm"] = np.array([wk["warmupIter"], float(wk["adaptH"]), wk["adaptHtarget"],
                                              float(wk["adaptDelta"]), wk["adaptDeltaTarget"],
                                              wk["adaptDeltaQuantile"]])
        pre = c["name"] + "__"
        out[pre + "cfg"] = np.array([c["target"], kind, kw["M"], kw["numIter"], c["minC"],
                                     c["maxC"], c["generated"], c["gen_k"], chain], dtype=np.int64)
        out[pre + "par"] = np.array([kw["H0"], 0.2, kw["delta0"], 2.0 / 3.0])
        out[pre + "q0"] = c["q0"]
        if c["tpar"] is not None:
            out[pre + "tpar"] = np.asarray(c["tpar"], float)
        for k in ("samples", "diagnostics", "trace", "tape_u", "tape_z", "n_u", "orbitMin", "orbitMax"):
            if k in res:
                out[pre + k] = res[k]
        d6 = res["diagnostics"]
        print(f"{c['name']:22s} calls={len(res['trace']):5d} evals={int(d6[:,6].sum()+d6[:,7].sum()):6d} "
              f"stop={sorted(set(d6[:,19].astype(int)))} depth~{d6[:,20].mean():.2f}")
    out["case_names"] = np.array([c["name"] for c in cases])
    out["seed"] = np.array([SEED], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "walnuts_traces.npz"), **out)

    # ---- G1: integrator-level known-answer tests ------------------------------------------
    g1 = {}
    krng = np.random.default_rng(11)
    kats = []
    A6, S6 = dense_precision(6, 50.0, 5)
    targets = [("std3", ol.T_STDGAUSS, td.stdGauss, None, 3), ("std100", ol.T_STDGAUSS, td.stdGauss, None, 100),
               ("funnel", ol.T_FUNNEL10, td.funnel10, None, 11),
               ("dense6", ol.T_DENSEGAUSS, make_dense(A6), A6.ravel(), 6)]
    def one_kat(tname, tk, lpF, d, ig, rep, krng):
        fn, kind = INTEGRATORS[ig]
        q = krng.normal(size=d)
        if tname == "funnel":
            q[0] = krng.normal() * 2.0
            q[1:] *= np.exp(0.5 * q[0])
        v = krng.normal(size=d)
        lp0, g = lpF(q)
        g = np.asarray(g, float)
        Ham0 = -lp0 + 0.5 * sum(v * v)
        h = float(krng.uniform(0.2, 1.5))
        xi = 1 if rep % 2 == 0 else -1
        delta = float(krng.choice([0.3, 0.05, 1e-3]))
        u = float(krng.uniform())
        minC = 0 if rep < 4 else 2
        aux = ai.integratorAuxPar(minC=minC, maxC=10 if rep != 3 else 3)
        saved = np.random.uniform
        np.random.uniform = lambda *a, **k: np.float64(u)
        try:
            o = fn(q, v, g, Ham0, h, xi, lpF, delta, aux)
        finally:
            np.random.uniform = saved
        kats.append(dict(t=tname, tk=tk, kind=kind, d=d, q=q, v=v, g=g, Ham0=Ham0, h=h,
                         xi=xi, delta=delta, u=u, minC=aux.minC, maxC=aux.maxC,
                         q1=np.asarray(o.q, float), v1=np.asarray(o.v, float),
                         g1=np.asarray(o.grad, float),
                         io=np.array([o.nEvalF, o.nEvalB, o.If, o.Ib, o.c]),
                         do=np.array([o.lp, o.lwt, o.igrConst])))

    for tname, tk, lpF, tpar, d in targets:
        for ig in ("fixed", "D", "R2P", "Y"):
            for rep in range(6):
                one_kat(tname, tk, lpF, d, ig, rep, krng)
    frng = np.random.default_rng(23)  # FlowD KATs from their own stream: the earlier ones keep their values
    for tname, tk, lpF, tpar, d in targets:
        for rep in range(6):
            one_kat(tname, tk, lpF, d, "F", rep, frng)
    g1["n"] = np.array([len(kats)])
    g1["dense6_A"] = A6.ravel()
    for i, k in enumerate(kats):
        p = f"k{i}__"
        g1[p + "meta"] = np.array([k["tk"], k["kind"], k["d"], k["xi"], k["minC"], k["maxC"]])
        g1[p + "par"] = np.array([k["Ham0"], k["h"], k["delta"], k["u"]])
        for f in ("q", "v", "g", "q1", "v1", "g1", "io", "do"):
            g1[p + f] = k[f]
    # lpFun values at random points
    pts = {}
    for tname, tk, lpF, tpar, d in targets + [("corr", ol.T_CORRGAUSS, td.corrGauss, None, 2),
                                              ("smile", ol.T_SMILE, td.smileDistr, None, 2),
                                              ("funnel1", ol.T_FUNNEL1, td.funnel1, None, 2),
                                              ("modfunnel", ol.T_MODFUNNEL, td.modFunnel, None, 2)]:
        Q = krng.normal(size=(8, d)) * 1.5
        L = []; G = []
        for q in Q:
            lp, g = lpF(q)
            L.append(float(lp)); G.append(np.asarray(g, float))
        pts[f"lp__{tname}__kind"] = np.array([tk, d])
        pts[f"lp__{tname}__q"] = Q
        pts[f"lp__{tname}__lp"] = np.array(L)
        pts[f"lp__{tname}__g"] = np.array(G)
    g1.update(pts)
    # G3: subTreePlan(2**i), i <= 6
    for i in range(0, 7):
        g1[f"plan__{i}"] = np.array(wn.subTreePlan(2 ** i), dtype=np.int64).reshape(-1, 2)
    # stopCondition
    sc = []
    for _ in range(32):
        d = int(krng.integers(1, 12))
        a = [krng.normal(size=d) for _ in range(4)]
        sc.append((a, bool(wn.stopCondition(*a))))
    g1["stop__n"] = np.array([len(sc)])
    for i, (a, r) in enumerate(sc):
        g1[f"stop__{i}__args"] = np.array(a)
        g1[f"stop__{i}__res"] = np.array([int(r)])
    # G5: P2quantile push/quantile sequences and np.quantile values (for 8f rank 1)
    import P2quantile as P2
    for k, prob in enumerate((0.2, 0.5, 0.9)):
        xs = krng.normal(size=300) * np.exp(krng.normal(size=300))
        f = P2.P2quantile(prob)
        qs = []
        for x in xs:
            f.push(x); qs.append(f.quantile())
        g1[f"p2__{k}__prob"] = np.array([prob]); g1[f"p2__{k}__x"] = xs; g1[f"p2__{k}__q"] = np.array(qs)
    qv = []
    for n in (1, 2, 3, 11, 12, 50, 333):
        for q in (0.9, 0.8, 0.5):
            a = krng.normal(size=n) ** 2
            qv.append((a, q, float(np.quantile(a, q))))
    g1["quant__n"] = np.array([len(qv)])
    for i, (a, q, r) in enumerate(qv):
        g1[f"quant__{i}__a"] = a; g1[f"quant__{i}__qr"] = np.array([q, r])
    np.savez_compressed(os.path.join(HERE, "integrator_kats.npz"), **g1)
    print("wrote", len(cases), "trace cases,", len(kats), "integrator KATs")


if __name__ == "__main__":
    main()
