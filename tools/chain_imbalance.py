"""Diagnostic: spread of leapfrog steps per chain in the headline workload (a launch lasts as long as its
slowest 16-chain group; measured mean/max = 0.91)."""
import sys, numpy as np
sys.path.insert(0, ".")
import torch, walnuts_amd as wa
from walnuts_amd import _lib as L
from bench import make_dense_problem, SEED
C, D = 4096, 100
A, q0 = make_dense_problem(D, 1e4, C, SEED)
e = wa.Engine(D, C, integrator=L.I_FIXED, H0=0.5, M=10, generated=L.G_Q0_SUMSQ).set_target(L.T_DENSEGAUSS, A)
s, d = e.run(torch.as_tensor(q0, device="cuda"), SEED, 100)
d = d.cpu().numpy()
steps = d[:, :, 6] + d[:, :, 7]
# ticks per iteration: steps + alignment wait (start at T = 1 mod 8)
per_chain = steps.sum(1)
print("steps/chain mean %.0f max %.0f min %.0f" % (per_chain.mean(), per_chain.max(), per_chain.min()))
g = per_chain.reshape(-1, 16)
print("per-WG max: mean of max %.0f, overall max %.0f; mean/overall max = %.3f; mean/mean-of-WG-max = %.3f" % (g.max(1).mean(), g.max(), per_chain.mean() / g.max(), per_chain.mean() / g.max(1).mean()))
print("kernel ms", e.last_run_ms(), "ticks if 12.7k cyc")
