import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle_lib as ol
from test_gpu_parity import dense_problem, run_engine, SEED
import walnuts_amd as wa
D, C = 200, 70
A, q0 = dense_problem(D, 1e4, C, 6)
c = dict(target=ol.T_DENSEGAUSS, integrator=ol.I_FIXED, H0=0.5, s=0.2, delta0=0.05, M=4, numIter=2, minC=0, maxC=10,
         R2Pprob0=2/3, generated=0, gen_k=0, tpar=A.ravel())
out = run_engine(wa, c, q0, 64, chain_offset=9000)
for k in (0, 69):
    o = ol.walnuts(ol.T_DENSEGAUSS, q0[k], integrator=ol.I_FIXED, tpar=A.ravel(), H0=0.5, numIter=2, M=4, seed=SEED, chain=9000 + k, trace_cap=64)
    n = min(o["ncalls"], out["trace_n"][k])
    print("chain", k, "calls", out["trace_n"][k], o["ncalls"])
    for r in range(min(n, 6)):
        print(" gpu", out["trace"][k, r, :6], out["trace"][k, r, 12:15])
        print(" ora", o["trace"][r, :6], o["trace"][r, 12:15])
for k in (0, 1, 69):
    u, z = ol.rng_tape(SEED, 9000 + k, 1, 4, D)
    lp0 = -0.5 * q0[k] @ A @ q0[k]
    print("chain", k, "lp0", lp0, "ke", 0.5 * (z * z).sum(), "H0", -lp0 + 0.5 * (z * z).sum(), "gpu Ham0 row0", out["trace"][k, 0, 4])
