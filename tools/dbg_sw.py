import sys, os, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle_lib as ol
from test_gpu_parity import run_engine, SEED
import walnuts_amd as wa
sw = np.load("tests/golden/stockwatson_data.npz"); T = int(sw["T"][0]); tpar = np.concatenate([[float(T)], sw["y"]])
D = 3 * T
q0 = np.zeros((1, D))
for ig in (ol.I_ADAPT_D, ol.I_ADAPT_FLOWD):
    c = dict(target=ol.T_STOCKWATSON, integrator=ig, H0=0.05, s=0.2, delta0=0.01, M=3, numIter=1, minC=0, maxC=10,
             R2Pprob0=2/3, generated=0, gen_k=0, tpar=tpar)
    out = run_engine(wa, c, q0, 64, chain_offset=500)
    o = ol.walnuts(ol.T_STOCKWATSON, q0[0], integrator=ig, tpar=tpar, H0=0.05, delta0=0.01, numIter=1, M=3, seed=SEED, chain=500, trace_cap=64)
    n = out["trace_n"][0]
    print("integrator", ig, "calls", n, o["ncalls"], out["kernel"])
    for r in range(min(n, 4)):
        print(" gpu", np.round(out["trace"][0, r, :12], 6))
        print(" ora", np.round(o["trace"][r, :12], 6))
# KAT through wn_integrator_step for SW + FlowD
rng = np.random.default_rng(1)
q = 0.1 * rng.normal(size=D); v = rng.normal(size=D)
lp0, g = ol.lpfun(ol.T_STOCKWATSON, q, tpar)
Ham0 = -lp0 + 0.5 * float(v @ v)
for ig in (ol.I_ADAPT_D, ol.I_ADAPT_FLOWD):
    e = wa.Engine(D, 1, integrator=ig, delta0=0.01, minC=0, maxC=10).set_target(ol.T_STOCKWATSON, tpar)
    q1, v1, g1, io, do = e.integrator_step(q[None], v[None], g[None], [Ham0], [0.05], [1])
    o = ol.integrator(ig, ol.T_STOCKWATSON, q, v, g, Ham0, 0.05, 1, 0.01, tpar=tpar)
    print("KAT", ig, "gpu io", io.cpu().numpy()[0], "do", do.cpu().numpy()[0], "| oracle", o["nEvalF"], o["nEvalB"], o["If"], o["Ib"], o["lp"], o["lwt"])
    e.close()
