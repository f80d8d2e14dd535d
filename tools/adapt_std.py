"""The adaptive integrators on std / diagonal Gaussians of 16 .. 128 dimensions (the reference's Gaussian scaling scripts, mainGaussESS.py):
gradient evaluations/s, one chain per wavefront (default: the trajectory-first kernel, wide mode) against the cooperative layout (route DIAG_COOP).  python tools/adapt_std.py   (GPU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import walnuts_amd as wa
from walnuts_amd import _lib as L
C = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
for D in ([int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else (32, 64, 100, 128)):
    q = torch.as_tensor(np.random.default_rng(1).normal(size=(C, D)), device=dev)
    for name, ig, kw in (("adaptLeapFrogD", L.I_ADAPT_D, dict(H0=1.4 * D ** -0.25, delta0=0.3)), ("adaptLeapFrogR2P", L.I_ADAPT_R2P, dict(H0=1.4 * D ** -0.25, delta0=0.3))):
        for env in ({}, {"WN_DIAG_COOP": "1"}):
            e = wa.Engine(D, C, integrator=ig, M=10, route=[k[3:] for k in env], **kw).set_target(L.T_STDGAUSS, None)
            e.run(q, 1, 20)
            s, d = e.run(q, 2, 20)
            ev = float((d[:, :, 6].sum() + d[:, :, 7].sum()).item())
            ms = e.last_run_ms()
            print("D=%-4d %-18s %-44s %.4g evaluations/s (%.1f ms)" % (D, name, e.kernel_name, ev / (ms * 1e-3), ms))
            e.close()
