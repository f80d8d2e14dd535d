#!/usr/bin/env python3
"""Time the UNMODIFIED Python reference (WALNUTSpy.WALNUTS) on BASELINE configs 2 and 3 in the build container
(BASELINE.md section 3, step 1).  The reference never travels to the GPU box: the number is written to
profiles/r02_python_reference_cpu.json and bench.py quotes it, labelled as measured here.

  MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python3 -B tools/ref_cpu_baseline.py
"""
import json
import multiprocessing as mp
import os
import platform
import sys
import time

sys.dont_write_bytecode = True
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/WALNUTSpy"


def _dense_chain(args):
    chain, iters = args
    sys.path.insert(0, REF)
    import WALNUTS as wn, adaptiveIntegrators as ai  # noqa: E401
    from bench import make_dense_problem, SEED
    A, q0 = make_dense_problem(100, 1e4, 8, SEED)

    def lp(q, hessian=False):
        g = -(A @ q)
        return [0.5 * np.dot(q, g), g]

    np.random.seed(1000 + chain)
    t0 = time.perf_counter()
    s, d = wn.WALNUTS(lp, q0[chain].copy(), integrator=ai.fixedLeapFrog, H0=0.5, numIter=iters, warmupIter=0, M=10)
    dt = time.perf_counter() - t0
    return float(d[:, 6].sum() + d[:, 7].sum()), dt


def _funnel_chain(args):
    chain, iters = args
    sys.path.insert(0, REF)
    import WALNUTS as wn, adaptiveIntegrators as ai, targetDistr as td  # noqa: E401
    rng = np.random.default_rng(chain)
    q0 = np.zeros(11); q0[0] = 3.0 * rng.normal(); q0[1:] = np.exp(0.5 * q0[0]) * rng.normal(size=10)
    np.random.seed(2000 + chain)
    t0 = time.perf_counter()
    s, d = wn.WALNUTS(td.funnel10, q0, integrator=ai.adaptLeapFrogR2P, H0=0.3, delta0=0.3, numIter=iters, warmupIter=0, M=12)
    dt = time.perf_counter() - t0
    return float(d[:, 6].sum() + d[:, 7].sum()), dt


def measure(fn, iters, cores):
    with mp.get_context("fork").Pool(cores) as pool:
        t0 = time.perf_counter()
        res = pool.map(fn, [(c, iters) for c in range(cores)], chunksize=1)
        wall = time.perf_counter() - t0
    steps = sum(r[0] for r in res)
    # rate of the sampling calls themselves (interpreter start-up and imports excluded): sum over the concurrent workers
    return dict(value=float(sum(r[0] / r[1] for r in res)), per_core=float(np.mean([r[0] / r[1] for r in res])), cores=cores, chains=cores,
                iterations=iters, steps=steps, wall_s_incl_startup=wall)


if __name__ == "__main__":
    import contextlib, io
    cores = os.cpu_count() or 1
    out = {"where": "build container (no GPU): %s, %d cores, Python %s, NumPy %s" % (platform.processor() or platform.machine(), cores,
                                                                                    platform.python_version(), np.__version__),
           "code": "unmodified /root/reference/WALNUTSpy (WALNUTS.py, adaptiveIntegrators.py, targetDistr.py), one chain per process"}
    with contextlib.redirect_stdout(io.StringIO()):
        one = measure(_dense_chain, 60, 1)
        allc = measure(_dense_chain, 60, cores)
        f1 = measure(_funnel_chain, 100, 1)
        fall = measure(_funnel_chain, 100, cores)
    out["cfg2_dense100_fixedLeapFrog"] = {"unit": "leapfrog steps/s", "one_core": one, "all_cores": allc}
    out["cfg3_funnel10_R2P"] = {"unit": "gradient evaluations/s", "one_core": f1, "all_cores": fall}
    json.dump(out, open(os.path.join(ROOT, "profiles", "r02_python_reference_cpu.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
