import sys, os, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import oracle_lib as ol
import walnuts_amd as wa
from test_gpu_orbit import funnel_q0, SEED, run_engine, same, CASES, INT_COLS
for case in CASES:
    integ, M, H0, delta0, minC, iters, gen, gk, rtol = case
    C=384
    q0 = funnel_q0(C, seed=11 + M)
    kw = dict(integrator=integ, H0=H0, delta0=delta0, M=M, minC=minC, generated=gen, gen_k=gk)
    sn, dn, _ = run_engine(wa, ol.T_FUNNEL10, q0, iters, **kw)
    so, do, _ = run_engine(wa, ol.T_FUNNEL10, q0, iters, legacy=True, **kw)
    ib = np.argwhere(dn[:,:,INT_COLS] != do[:,:,INT_COLS])
    rel = np.abs(sn-so)/np.maximum(1e-300, np.maximum(np.abs(so), 1e-3))
    print(case, 'int mismatches', len(ib), 'chains', len(np.unique(ib[:,0])) if len(ib) else 0, 'first iters', ib[:3].tolist(),
          'draw rel diff: it1 %.2e  last %.2e  max %.2e' % (rel[:,1].max(), rel[:,-1].max(), rel.max()), 'nbits-differ chains', int((np.abs(sn-so).max(axis=(1,2))>0).sum()))

for target in (ol.T_CORRGAUSS, ol.T_SMILE, ol.T_FUNNEL1, ol.T_MODFUNNEL):
    C, iters, M = 128, 25, 7
    q0 = np.random.default_rng(9).normal(size=(C, 2))
    kw = dict(integrator=ol.I_ADAPT_R2P, H0=0.4, delta0=0.2, M=M)
    sn, dn, _ = run_engine(wa, target, q0, iters, **kw)
    so, do, _ = run_engine(wa, target, q0, iters, legacy=True, **kw)
    ib = np.argwhere(dn[:,:,INT_COLS] != do[:,:,INT_COLS])
    rel = np.abs(sn-so)/np.maximum(1e-300, np.maximum(np.abs(so), 1e-3))
    print('target', target, 'int mismatches', len(ib), 'first', ib[:3].tolist(), 'draw rel diff: it1 %.2e it5 %.2e last %.2e' % (rel[:,1].max(), rel[:,5].max(), rel[:,-1].max()))
