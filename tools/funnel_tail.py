"""Where the funnel launch (SURVEY 8d cfg 3, bench.py --workload funnel) spends its time: runs the 4096-chain launch, finds
the chain with the most gradient evaluations, and runs THAT chain alone -- one chain per wavefront (the product path)
and one chain per lane.  Measured (round 2): launch 744 ms, busiest chain alone 708 ms (1.73 us per evaluation); the
launch is one chain's serial latency, not throughput.  Run on the GPU box: python tools/funnel_tail.py"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import walnuts_amd as wa
from walnuts_amd import _lib as L
SEED = 20251121
C = 4096
rng = np.random.default_rng([SEED, 3, 0])
q0 = np.zeros((C, 11)); q0[:, 0] = 3.0 * rng.normal(size=C)
q0[:, 1:] = np.exp(0.5 * q0[:, :1]) * rng.normal(size=(C, 10))
dev = torch.device('cuda:0')
def run(C_, off, q, wpg, iters=300):
    e = wa.Engine(11, C_, integrator=L.I_ADAPT_R2P, H0=0.3, delta0=0.3, M=12, generated=L.G_FIRSTK, gen_k=2, chain_offset=off, device=0, waves_per_group=wpg)
    e.set_target(L.T_FUNNEL10)
    s, d = e.run(torch.as_tensor(q, device=dev), SEED, iters)
    s, d = e.run(torch.as_tensor(q, device=dev), SEED, iters)
    ms = e.last_run_ms(); name = e.kernel_name
    ev = (d[:, :, 6] + d[:, :, 7]).sum(1).cpu().numpy(); steps = d[:, :, 2].sum(1).cpu().numpy()
    e.close()
    return ms, ev, name, d.cpu().numpy()
ms, ev, name, d = run(C, 0, q0, 0)
k = int(ev.argmax())
print("full", name, ms, "busiest", k, ev[k], "q0[0]", q0[k, 0])
srt = np.sort(ev)[::-1]
print("top evals", srt[:12], "median", np.median(ev), "sum", ev.sum())
for wpg in (0, 64):
    ms1, ev1, name1, d1 = run(1, k, q0[k:k+1], wpg)
    print(name1, "1 chain ms", ms1, "evals", ev1[0], "us/eval", ms1 * 1e3 / ev1[0])
