"""Cost distribution of one transition of the reference sampler walnuts/walnuts.py (gradient evaluations per walnuts_step call) on the 10-D
standard normal with test/test.py parameters: run in the build container only (imports /root/reference).  profiles/r04_walnuts_tail.txt."""
import importlib.util, sys
sys.dont_write_bytecode = True  # nothing is ever written into /root/reference (round 4 left a __pycache__ there: reported in DESIGN.md)
import numpy as np, time
spec = importlib.util.spec_from_file_location("walnuts_ref", "/root/reference/walnuts/walnuts.py")
w = importlib.util.module_from_spec(spec); spec.loader.exec_module(w)
D=10
cnt=[0]
def logp(th): return -0.5*np.dot(th,th)
def grad(th):
    cnt[0]+=1
    return -th
res=[]
t0=time.time()
for seed in range(8):
    rng=np.random.default_rng(1000+seed)
    th=np.zeros(D)
    for it in range(1500):
        c0=cnt[0]
        th = w.walnuts_step(rng, th, logp, grad, np.ones(D), 2.0, 10, 0.1)
        res.append(cnt[0]-c0)
r=np.array(res)
print('transitions',len(r),'sec',time.time()-t0)
print('grad evals per transition: mean %.1f median %.1f p90 %.0f p99 %.0f p99.9 %.0f max %d'%(r.mean(),np.median(r),np.percentile(r,90),np.percentile(r,99),np.percentile(r,99.9),r.max()))
print('share of all evals in the top 1%% of transitions: %.2f'%(np.sort(r)[-len(r)//100:].sum()/r.sum()))
