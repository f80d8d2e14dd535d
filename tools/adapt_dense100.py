"""adaptLeapFrogD / adaptLeapFrogR2P on BASELINE configs[1]'s geometry (100-D dense Gaussian, kappa 1e4): gradient evaluations/s of the
cooperative MFMA-layout kernel (these integrators have no role-split kernel).  python tools/adapt_dense100.py [chains]   (GPU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import walnuts_amd as wa
from walnuts_amd import _lib as L
from bench import make_dense_problem
C = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
A, q0 = make_dense_problem(100, 1e4, C, 20251121)
dev = torch.device("cuda", 0)
q = torch.as_tensor(q0, device=dev)
for name, ig, kw in (("fixedLeapFrog", L.I_FIXED, dict(H0=0.5)), ("adaptLeapFrogD", L.I_ADAPT_D, dict(H0=0.5, delta0=0.3)),
                     ("adaptLeapFrogR2P", L.I_ADAPT_R2P, dict(H0=0.5, delta0=0.3)), ("adaptLeapFrogFlowD", L.I_ADAPT_FLOWD, dict(H0=0.5, delta0=0.05))):
    e = wa.Engine(100, C, integrator=ig, M=10, **kw).set_target(L.T_DENSEGAUSS, A)
    e.run(q, 1, 20)
    s, d = e.run(q, 2, 20)
    ev = float((d[:, :, 6].sum() + d[:, :, 7].sum()).item())
    ms = e.last_run_ms()
    print("%-20s %-52s %.4g gradient evaluations/s = %.1f TFLOP/s (%.1f ms)" % (name, e.kernel_name, ev / (ms * 1e-3), ev * 21200 / (ms * 1e-3) / 1e12, ms))
    e.close()
