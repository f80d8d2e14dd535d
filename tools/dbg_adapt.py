import sys, numpy as np
sys.path.insert(0, ".")
import torch, walnuts_amd as wa
from walnuts_amd import _lib as L
from bench import make_dense_problem, SEED
D = 100
A, q0 = make_dense_problem(D, 1e4, 4096, SEED)
for tgt, par, name in ((L.T_DENSEGAUSS, A, "dense100"), (L.T_STDGAUSS, None, "std100")):
    for ig, iname in ((L.I_ADAPT_R2P, "R2P"), (L.I_ADAPT_D, "D")):
        for C in (1024, 2048, 4096):
            e = wa.Engine(D, C, integrator=ig, H0=1.2 if tgt == L.T_DENSEGAUSS else 1.4 * D ** -0.25, delta0=0.3, M=8, generated=L.G_Q0_SUMSQ).set_target(tgt, par)
            q = torch.as_tensor(q0[:C] if tgt == L.T_DENSEGAUSS else np.random.default_rng(1).normal(size=(C, D)), device="cuda")
            e.run(q, 1, 20)
            s, d = e.run(q, 2, 20)
            ev = float((d[:, :, 6] + d[:, :, 7]).sum())
            ms = e.last_run_ms()
            print(f"{name} {iname} C={C}: {ev/ms*1e3:.3e} evals/s, {ms:.1f} ms, kernel {e.kernel_name}")
            e.close()
