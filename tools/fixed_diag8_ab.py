"""fixedLeapFrog on std Gaussians of 200 / 256 / 512 dimensions: the trajectory-first kernel's wide mode (default up to 256) against the cooperative layout / the wave layout's engine (route DIAG_COOP).  (GPU)"""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import walnuts_amd as wa
from walnuts_amd import _lib as L
C = 4096; dev = torch.device("cuda", 0)
for D in (200, 256, 512):
    q = torch.as_tensor(np.random.default_rng(1).normal(size=(C, D)), device=dev)
    for env in ({}, {"WN_DIAG_COOP": "1"}):
        e = wa.Engine(D, C, integrator=L.I_FIXED, M=10, H0=1.4 * D ** -0.25, route=[k[3:] for k in env]).set_target(L.T_STDGAUSS, None)
        e.run(q, 1, 20); s, d = e.run(q, 2, 20)
        ev = float((d[:, :, 6].sum() + d[:, :, 7].sum()).item()); ms = e.last_run_ms()
        print("D=%-4d fixed %-44s %.4g steps/s (%.1f ms)" % (D, e.kernel_name, ev / (ms * 1e-3), ms)); e.close()
