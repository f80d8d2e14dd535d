"""Diagnostic: leapfrog steps/s of the cluster layout (std Gaussian, D up to 2^18) and its nominal HBM figure
(48*D bytes per gradient evaluation, the unfused streaming formulation)."""
import sys, numpy as np
sys.path.insert(0, ".")
import torch, walnuts_amd as wa
from walnuts_amd import _lib as L
for D, C in ((8192, 64), (65536, 16), (262144, 4), (262144, 1)):
    rng = np.random.default_rng(D)
    q0 = rng.normal(size=(C, D))
    e = wa.Engine(D, C, integrator=L.I_FIXED, H0=1.4 * D ** -0.25, M=10, generated=L.G_Q0_SUMSQ).set_target(L.T_STDGAUSS)
    qd = torch.as_tensor(q0, device="cuda")
    e.run(qd, 1, 2)
    s, d = e.run(qd, 2, 20)
    st = float((d[:, :, 6] + d[:, :, 7]).sum()); ms = e.last_run_ms()
    print(f"D={D} C={C} {e.kernel_name}: {st/ms*1e3:.3e} steps/s, tick {ms*1e3/(st/C):.2f} us, "
          f"nominal {st/ms*1e3*48*D/1e9:.0f} GB/s, E[sum q^2]/D = {s[:, 5:, 1].mean().item()/D:.4f}")
    e.close()
try:
    wa.Engine(262144, 8).set_target(L.T_STDGAUSS)
except NotImplementedError as ex:
    print("too many chains ->", ex)
