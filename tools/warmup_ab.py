"""The reference's default call shape (fixedLeapFrog + warm-up adaptation of H and delta, WALNUTS.py:111-125) on BASELINE configs[1]'s
geometry: leapfrog steps/s of the headline kernel (round 5: warm-up runs on wn_fast4_kernel) against the one-stream kernel of the same
layout (route COOP_LEGACY, where such runs went before).  python tools/warmup_ab.py [chains] [iters] [warmupIter]   (GPU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import walnuts_amd as wa
from walnuts_amd import _lib as L
from bench import make_dense_problem

C = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 50
A, q0 = make_dense_problem(100, 1e4, C, 20251121)
dev = torch.device("cuda", 0)
for legacy in (False, True):
    e = wa.Engine(100, C, integrator=L.I_FIXED, H0=0.5, M=10, warmupIter=warm, generated=L.G_Q0_SUMSQ, route="COOP_LEGACY" if legacy else 0).set_target(L.T_DENSEGAUSS, A)
    q = torch.as_tensor(q0, device=dev)
    e.run(q, 1, iters)
    s, d = e.run(q, 2, iters)
    steps = float((d[:, :, 6].sum() + d[:, :, 7].sum()).item())
    ms = e.last_run_ms()
    print("%-60s %.4g leapfrog steps/s  (%.1f ms, %d chains x %d iterations, warmupIter %d; H after warm-up: median %.4f)" %
          (e.kernel_name, steps / (ms * 1e-3), ms, C, iters, warm, float(d[:, -1, 15].median().item())))
    e.close()
