"""recordOrbitStats = True on funnel10 / adaptLeapFrogR2P (mainFunnel.py:36-41, mainFunnelOrbitLen.py, mainFunnelTransient.py): the trajectory-first
kernel against the one-stream engine (route ORBIT_LEGACY).  python tools/orbit_stats_ab.py   (GPU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import walnuts_amd as wa
from walnuts_amd import _lib as L
C = 4096; dev = torch.device("cuda", 0)
rng = np.random.default_rng(3)
qf = np.zeros((C, 11)); qf[:, 0] = 3.0 * rng.normal(size=C); qf[:, 1:] = np.exp(0.5 * qf[:, :1]) * rng.normal(size=(C, 10))
q = torch.as_tensor(qf, device=dev)
for env in ({}, {"WN_ORBIT_LEGACY": "1"}):
    e = wa.Engine(11, C, integrator=L.I_ADAPT_R2P, H0=0.3, delta0=0.3, M=12, generated=L.G_FIRSTK, gen_k=2, route=[k[3:] for k in env]).set_target(L.T_FUNNEL10, None)
    e.run_orbit_stats(q, 1, 100)
    s, d, mn, mx = e.run_orbit_stats(q, 2, 100)
    ev = float((d[:, :, 6].sum() + d[:, :, 7].sum()).item()); ms = e.last_run_ms()
    print("funnel10 R2P M=12, 100 iterations, orbit statistics: %-44s %.4g evaluations/s (%.1f ms)" % (e.kernel_name, ev / (ms * 1e-3), ms))
    e.close()
