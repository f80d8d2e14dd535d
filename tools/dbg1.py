import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle_lib as ol, walnuts_amd as wa
from walnuts_amd import _lib as L
rng = np.random.default_rng(1)
q0 = np.r_[-2.0, np.exp(-1.0) * rng.normal(size=10)]
for ig in (0, 1, 2):
    e = wa.Engine(11, 1, integrator=ig, H0=0.3, delta0=0.3, M=5).set_target(L.T_FUNNEL10)
    s, d, tr, tn = e.run(q0[None], 5, 2, trace_cap=200)
    tr = tr.cpu().numpy()[0]; n = int(tn.cpu()[0])
    o = ol.walnuts(ol.T_FUNNEL10, q0, integrator=ig, H0=0.3, delta0=0.3, numIter=2, M=5, seed=5, chain=0, trace_cap=200)
    print("integrator", ig, "calls gpu/oracle", n, o["ncalls"])
    for r in range(3):
        print("  gpu", np.array2string(tr[r, :12], precision=5), "\n  ora", np.array2string(o["trace"][r, :12], precision=5))
    lp, g = e.lpfun(q0[None]); print("  lpfun", float(lp.cpu()[0]), ol.lpfun(ol.T_FUNNEL10, q0)[0])
