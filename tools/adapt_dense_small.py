"""adaptLeapFrogD / R2P on dense Gaussians of 24 .. 64 dimensions: the cooperative layout (route DENSE_COOP) against one chain per wavefront with
the matrix in LDS.  python tools/adapt_dense_small.py   (GPU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import walnuts_amd as wa
from walnuts_amd import _lib as L
from bench import make_dense_problem
C = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
for D in ([int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else (24, 40, 64)):
    A, q0 = make_dense_problem(D, 1e3, C, 20251121)
    q = torch.as_tensor(q0, device=dev)
    for name, ig in (("adaptLeapFrogD", L.I_ADAPT_D), ("adaptLeapFrogR2P", L.I_ADAPT_R2P)):
        for env in ({}, {"WN_DENSE_COOP": "1"}):
            e = wa.Engine(D, C, integrator=ig, M=10, H0=0.5, delta0=0.3, route=[k[3:] for k in env]).set_target(L.T_DENSEGAUSS, A)
            e.run(q, 1, 20)
            s, d = e.run(q, 2, 20)
            ev = float((d[:, :, 6].sum() + d[:, :, 7].sum()).item())
            ms = e.last_run_ms()
            print("D=%-4d %-18s %-48s %.4g evaluations/s (%.1f ms)" % (D, name, e.kernel_name, ev / (ms * 1e-3), ms))
            e.close()
