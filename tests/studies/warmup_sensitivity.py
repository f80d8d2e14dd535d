#!/usr/bin/env python3
"""How sensitive is a WARM-UP run of the reference's algorithm to the last bit of its arithmetic?  (Next #3.)

CPU only, oracle only (test infrastructure: this file lives under tests/).  Every chain is run twice through the oracle
(oracle/walnuts_oracle.c = the reference's WALNUTS.py:111-727, pinned by the goldens): once as is, once with ONE coordinate of q0
moved by one unit in the last place.  Same random stream, same everything else.  Printed per target: the fraction of chains whose
adapted step size H (diagnostics col 15) / delta (col 18) / draw still agree to 1e-6 and to 1e-4 after k iterations, and whether the
integer columns (6, 7, 19, 20) still agree.  This is the curve an implementation that differs from the reference ONLY in the order of
its floating-point sums (any GPU kernel; NumPy's pairwise `np.sum` against the builtin `sum` of adaptiveIntegrators.py:55) has to be
judged against.

  python tests/studies/warmup_sensitivity.py [chains] [iterations] [warmup]
"""
import os
import sys
from multiprocessing import Pool

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as ol  # noqa: E402

SEED = 20251121


def problems(C):
    from bench import make_dense_problem
    rng = np.random.default_rng(5)
    A, q0d = make_dense_problem(100, 1e4, C, SEED)
    qf = np.zeros((C, 11)); qf[:, 0] = 3.0 * rng.normal(size=C); qf[:, 1:] = np.exp(0.5 * qf[:, :1]) * rng.normal(size=(C, 10))
    return {
        "std100_fixed": dict(target=ol.T_STDGAUSS, tpar=None, q0=rng.normal(size=(C, 100)), kw=dict(integrator=ol.I_FIXED, H0=0.2, delta0=0.05, M=10)),
        "dense100_fixed": dict(target=ol.T_DENSEGAUSS, tpar=A.ravel(), q0=q0d, kw=dict(integrator=ol.I_FIXED, H0=0.5, delta0=0.05, M=10)),
        "funnel10_R2P": dict(target=ol.T_FUNNEL10, tpar=None, q0=qf, kw=dict(integrator=ol.I_ADAPT_R2P, H0=0.3, delta0=0.3, M=12)),
    }


def _one(a):
    target, tpar, q0, chain, iters, warm, kw = a
    o = ol.walnuts(target, q0, tpar=tpar, numIter=iters, warmupIter=warm, seed=SEED, chain=chain, **kw)
    return o["diagnostics"], o["samples"].T  # [iters + 1, dg]


def curve(da, sa, db, sb):
    """da, db [C, iters, 24]; sa, sb [C, iters+1, dg] -> per iteration: fraction of chains within 1e-6 / 1e-4 so far (H, delta, draw), ints equal so far"""
    rel = lambda x, y: np.abs(x - y) / np.maximum(np.abs(y), 1e-300)
    eH = np.maximum.accumulate(np.maximum(rel(da[..., 15], db[..., 15]), rel(da[..., 18], db[..., 18])), axis=1)
    eS = np.maximum.accumulate((np.abs(sa - sb) / (1e-9 / 1e-6 + np.abs(sb))).max(-1)[:, 1:], axis=1)
    ints = np.logical_and.accumulate((da[..., [6, 7, 19, 20]] == db[..., [6, 7, 19, 20]]).all(-1), axis=1)
    return {"H6": (eH <= 1e-6).mean(0), "H4": (eH <= 1e-4).mean(0), "S6": (eS <= 1e-6).mean(0), "S4": (eS <= 1e-4).mean(0), "int": ints.mean(0)}


def main():
    C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    warm = int(sys.argv[3]) if len(sys.argv) > 3 else iters
    marks = [k for k in (1, 2, 5, 10, 12, 20, 50, 100, 200, 500, 1000) if k <= iters]
    with Pool(os.cpu_count()) as pool:
        for name, p in problems(C).items():
            qb = p["q0"].copy(); qb[:, 0] = np.nextafter(qb[:, 0], np.inf)
            ra = pool.map(_one, [(p["target"], p["tpar"], p["q0"][c], c, iters, warm, p["kw"]) for c in range(C)])
            rb = pool.map(_one, [(p["target"], p["tpar"], qb[c], c, iters, warm, p["kw"]) for c in range(C)])
            cv = curve(np.stack([r[0] for r in ra]), np.stack([r[1] for r in ra]), np.stack([r[0] for r in rb]), np.stack([r[1] for r in rb]))
            print(f"{name}: {C} chains, {warm} warm-up iterations of {iters}; oracle vs oracle with q0[0] moved by 1 ulp")
            print("  after iteration      " + " ".join(f"{k:>6d}" for k in marks))
            for key, lab in (("H6", "H, delta within 1e-6"), ("H4", "H, delta within 1e-4"), ("S6", "draws within 1e-6   "), ("S4", "draws within 1e-4   "), ("int", "integer cols equal  ")):
                print(f"  {lab} " + " ".join(f"{cv[key][k - 1]:6.3f}" for k in marks))
            sys.stdout.flush()


if __name__ == "__main__":
    main()
