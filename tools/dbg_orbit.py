import sys, os, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import oracle_lib as ol
import walnuts_amd as wa
from test_gpu_orbit import funnel_q0, SEED, run_engine, same
q0 = funnel_q0(384, seed=23)
kw = dict(integrator=ol.I_ADAPT_R2P, H0=0.3, delta0=0.3, M=12)
sn, dn, _ = run_engine(wa, ol.T_FUNNEL10, q0, 40, **kw)
so, do, _ = run_engine(wa, ol.T_FUNNEL10, q0, 40, legacy=True, **kw)
bad = np.argwhere(~((dn == do) | (np.isnan(dn) & np.isnan(do))))
print('mismatches', len(bad), 'columns', np.unique(bad[:,2]) if len(bad) else None)
for c,it,k in bad[:12]:
    o = ol.walnuts(ol.T_FUNNEL10, q0[c], integrator=ol.I_ADAPT_R2P, H0=0.3, delta0=0.3, numIter=it+1, M=12, seed=SEED, chain=int(c))
    print(c,it,k, float(dn[c,it,k]).hex(), float(do[c,it,k]).hex(), float(o['diagnostics'][it,k]).hex(), 'stop', dn[c,it,19], 'Nd', dn[c,it,20], 'nevalF', dn[c,it,6])
