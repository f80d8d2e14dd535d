import os, sys, json, subprocess
sys.path.insert(0, '.')
import numpy as np, torch
import walnuts_amd as wa
from walnuts_amd import _lib as L
from bench import make_dense_problem, SEED
dev = torch.device('cuda', 0)
for C in (1024, 2048, 4096, 8192):
    A, q0 = make_dense_problem(100, 1e4, C, SEED)
    e = wa.Engine(100, C, integrator=L.I_FIXED, H0=0.5, M=10, generated=L.G_Q0_SUMSQ).set_target(L.T_DENSEGAUSS, A)
    q0d = torch.as_tensor(q0, device=dev)
    e.run(q0d, SEED + 7, 20)
    s, d = e.run(q0d, SEED, 100)
    st = float((d[:, :, 6] + d[:, :, 7]).sum().item()); ms = e.last_run_ms()
    print(f"C={C} {e.kernel_name}: {st / (ms * 1e-3):.4g} steps/s ({ms:.1f} ms)", flush=True)
    e.close()
