"""Randomised differential test of the wave-pair headline kernel (wn_k_fast4.hip) against the CPU oracle: random dimension (all four
geometries), chain count (1, 2 and 4 pairs per workgroup, ragged last groups), depth, step size, conditioning, generated kind,
traced or plain.  Every diagnostics column, every draw and -- traced -- every row of the per-step trace of the checked chains.
python tools/fuzz_fast4.py [n] [seed]   (GPU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
import walnuts_amd as wa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
INT = (0, 1, 4, 5, 6, 7, 8, 9, 12, 13, 19, 20, 21, 22)
bad = 0
other = 0
ONLY = int(os.environ.get("FUZZ_ONLY", "-1"))  # replay one configuration of a run (same n and seed), with details


def close(a, b, tol=1e-6):
    # (overflowing energies: the engine's dot products are FMA chains, the oracle's -- as the reference's -- separate multiplications
    # and additions; where a product overflows, fma(q, g, -inf) is -inf and q*g + -inf is NaN.  Both are "not finite", which is all the
    # sampler asks of them (stop code 999): non-finite values of either kind compare equal here.)
    with np.errstate(all="ignore"):
        scale = np.nanmax(np.abs(b), axis=0, keepdims=True) if b.size else 1.0
        return bool(((~np.isfinite(a) & ~np.isfinite(b)) |
                     (np.abs(a - b) <= 1e-9 + tol * np.maximum(np.abs(b), scale))).all())


for it in range(n):
    D = int(rng.choice([5, 16, 17, 31, 32, 33, 47, 64, 65, 80, 96, 97, 99, 100]))
    C = int(rng.choice([1, 3, 4, 5, 17, 70, 300, 1100, 2500, 4300], p=[.1, .1, .1, .1, .15, .15, .1, .1, .05, .05]))
    M = int(rng.integers(1, 9)) if rng.random() < 0.85 else int(rng.integers(9, 12))
    if C > 1000: M = min(M, 8)
    numIter = int(rng.integers(1, 6))
    Q, _ = np.linalg.qr(rng.normal(size=(D, D))); lam = np.logspace(0, rng.uniform(0.0, 4.0), D)
    A = (Q * (1.0 / lam)) @ Q.T; A = 0.5 * (A + A.T)
    q0 = rng.normal(size=(C, D)) @ (Q * np.sqrt(lam)).T
    if rng.random() < 0.1: q0[: max(1, C // 7)] *= 10.0 ** rng.uniform(100, 200)  # overflowing energies: stop code 999, zero / infinite weights
    H0 = float(rng.uniform(0.2, 1.1))
    gen, gk = (ol.G_IDENTITY, 0) if rng.random() < 0.4 else ((ol.G_Q0_SUMSQ, 0) if rng.random() < 0.5 else (ol.G_FIRSTK, int(rng.integers(1, min(3, D) + 1))))
    traced = rng.random() < 0.35 and C <= 300
    cap = numIter * (1 << M) + 8
    off = int(rng.integers(0, 1 << 20))
    if ONLY >= 0 and it != ONLY:  # (replay: the random stream is consumed as in a full run, nothing is run)
        rng.integers(0, C)
        continue
    e = wa.Engine(D, C, integrator=ol.I_FIXED, H0=H0, M=M, generated=gen, gen_k=gk, chain_offset=off)
    e.set_target(ol.T_DENSEGAUSS, A)
    name = e.kernel_name
    if "wn_fast4_kernel" not in name:  # (small dimensions with many chains go to the lane layout)
        other += 1; e.close(); continue
    out = e.run(q0, 4242 + it, numIter, trace_cap=cap if traced else 0)
    s = out[0].transpose(1, 2).cpu().numpy(); d = out[1].cpu().numpy()
    tr, tn = (out[2].cpu().numpy(), out[3].cpu().numpy()) if traced else (None, None)
    e.close()
    for k in sorted({0, C - 1, C // 2, min(C - 1, 3), min(C - 1, 4), int(rng.integers(0, C))}):
        o = ol.walnuts(ol.T_DENSEGAUSS, q0[k], integrator=ol.I_FIXED, tpar=A.ravel(), H0=H0, numIter=numIter, M=M, seed=4242 + it,
                       chain=off + k, generated=gen, gen_k=gk, trace_cap=cap if traced else 0)
        ok = all(np.array_equal(d[k][:, c], o["diagnostics"][:, c]) for c in INT) and close(d[k], o["diagnostics"]) and close(s[k], o["samples"])
        if traced and ok:
            m = o["ncalls"]
            ok = tn[k] == m and close(tr[k, :m], o["trace"][:m])
        if not ok and ONLY >= 0:
            dd, od = d[k], o["diagnostics"]
            print("  integer columns differing:", [c for c in INT if not np.array_equal(dd[:, c], od[:, c])])
            with np.errstate(all="ignore"):
                print("  max rel diff of diagnostics per column:", np.round(np.nanmax(np.abs(dd - od) / np.maximum(1e-300, np.abs(od)), axis=0), 12).tolist())
                print("  engine rows:\n", dd[:, [2, 6, 7, 17, 19, 20]], "\n  oracle rows:\n", od[:, [2, 6, 7, 17, 19, 20]])
                print("  draws max abs diff per iteration:", np.abs(s[k] - o["samples"]).max(axis=0).tolist())
        if not ok:
            bad += 1
            print(f"MISMATCH it={it} D={D} C={C} M={M} iters={numIter} gen={gen} H0={H0:.4f} traced={traced} chain={k} kernel={name}", flush=True)
print(f"fuzz_fast4: {n} configurations ({other} not on the pair kernel), {bad} mismatching chains")
