"""Runs WITH warm-up adaptation (the reference's mainFunnel.py:27-29 / example.py:25-27: adaptLeapFrogR2P, warmupIter = half of the run) on the
trajectory-first kernel (default) against the kernels they used before its WARM instantiations (routes ORBIT_LEGACY / DIAG_COOP / DENSE_COOP).
python tools/warm_orbit_ab.py   (GPU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import walnuts_amd as wa
from walnuts_amd import _lib as L
from bench import make_dense_problem
dev = torch.device("cuda", 0)
C = 4096
rng = np.random.default_rng(3)
qf = np.zeros((C, 11)); qf[:, 0] = 3.0 * rng.normal(size=C); qf[:, 1:] = np.exp(0.5 * qf[:, :1]) * rng.normal(size=(C, 10))
A, qd = make_dense_problem(100, 1e4, C, 20251121)
cases = [("funnel10 R2P M=12, 100 it (50 warm)", 11, L.T_FUNNEL10, None, qf, dict(integrator=L.I_ADAPT_R2P, H0=0.3, delta0=0.3, M=12, warmupIter=50), 100, {"WN_ORBIT_LEGACY": "1"}),
         ("std 100-D R2P M=10, 40 it (20 warm)", 100, L.T_STDGAUSS, None, rng.normal(size=(C, 100)), dict(integrator=L.I_ADAPT_R2P, H0=0.45, delta0=0.3, M=10, warmupIter=20), 40, {"WN_DIAG_COOP": "1"}),
         ("dense 100-D D M=10, 40 it (20 warm)", 100, L.T_DENSEGAUSS, A, qd, dict(integrator=L.I_ADAPT_D, H0=0.5, delta0=0.3, M=10, warmupIter=20), 40, {"WN_DENSE_COOP": "1"})]
for name, D, tk, tpar, q0, kw, it, env in cases:
    q = torch.as_tensor(q0, device=dev)
    for en in ({}, env):
        e = wa.Engine(D, C, route=[k[3:] for k in en], **kw).set_target(tk, tpar)
        e.run(q, 1, it)
        s, d = e.run(q, 2, it)
        ev = float((d[:, :, 6].sum() + d[:, :, 7].sum()).item())
        ms = e.last_run_ms()
        print("%-38s %-50s %.4g evaluations/s (%.1f ms)" % (name, e.kernel_name, ev / (ms * 1e-3), ms))
        e.close()
