"""adaptLeapFrogFlowD (adaptiveIntegrators.py:246-356) on Gaussians: the cooperative MFMA layout (round 5) against where such runs went before
(one chain per wavefront for std / diagonal, split mode for dense; WN_FLOW_NOCOOP=1).  gradient evaluations/s.  python tools/flowd_ab.py  (GPU)"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, torch
    import walnuts_amd as wa
    from walnuts_amd import _lib as L
    from bench import make_dense_problem
    kind, D, C = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    dev = torch.device("cuda", 0)
    if kind == "dense":
        A, q0 = make_dense_problem(D, 1e2, C, 7); tk, tp = L.T_DENSEGAUSS, A
    else:
        q0 = np.random.default_rng(1).normal(size=(C, D)); tk, tp = L.T_STDGAUSS, None
    e = wa.Engine(D, C, integrator=L.I_ADAPT_FLOWD, H0=0.6, delta0=0.05, M=8, route=sys.argv[5].split(",") if len(sys.argv) > 5 and sys.argv[5] else 0).set_target(tk, tp)
    q = torch.as_tensor(q0, device=dev)
    e.run(q, 1, 10)
    s, d = e.run(q, 2, 10)
    ev = float((d[:, :, 6].sum() + d[:, :, 7].sum()).item())
    print(json.dumps(dict(kernel=e.kernel_name, evals_per_s=ev / (e.last_run_ms() * 1e-3), ms=e.last_run_ms())))
    sys.exit(0)
for kind, D, C in (("dense", 40, 4096), ("dense", 100, 4096), ("std", 60, 4096), ("std", 100, 4096), ("dense", 100, 1024)):
    for env in ({}, {"WN_FLOW_NOCOOP": "1"}):
        r = subprocess.run([sys.executable, __file__, "child", kind, str(D), str(C), ",".join(k[3:] for k in env)], capture_output=True, text=True)
        try:
            j = json.loads(r.stdout.strip().splitlines()[-1])
            print("%-6s D=%-4d C=%-5d %-62s %.4g evaluations/s (%.1f ms)" % (kind, D, C, j["kernel"], j["evals_per_s"], j["ms"]))
        except Exception:
            print(kind, D, C, env, "FAILED", r.stderr[-300:])
