"""Randomised differential test: GPU engine vs CPU oracle over random (target, D, integrator, M, step size,
warm-up, generated, chain count) configurations.  python tools/fuzz_parity.py [n] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
import walnuts_amd as wa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
QBITS = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # adapt_quant_bits of the warm-up runs, engine and oracle alike (0 = the reference's arithmetic)
sw = np.load(os.path.join(ROOT, "tests", "golden", "stockwatson_data.npz"))
SWT = int(sw["T"][0]); SWP = np.concatenate([[float(SWT)], sw["y"]])
bad = 0
for it in range(n):
    tk = int(rng.choice([ol.T_STDGAUSS, ol.T_DIAGGAUSS, ol.T_DENSEGAUSS, ol.T_FUNNEL10, ol.T_CORRGAUSS, ol.T_SMILE,
                         ol.T_FUNNEL1, ol.T_MODFUNNEL, ol.T_STOCKWATSON], p=[.2, .2, .25, .1, .04, .04, .04, .04, .09]))
    ig = int(rng.integers(0, 5))
    C = int(rng.choice([1, 3, 17, 70, 8192], p=[.24, .24, .24, .24, .04]))
    M = int(rng.integers(2, 8)) if rng.random() < 0.9 else int(rng.integers(8, 11))
    numIter = int(rng.integers(1, 6))
    warm = int(rng.choice([0, 0, 4]))
    tpar = None
    if tk in (ol.T_STDGAUSS, ol.T_DIAGGAUSS):
        D = int(rng.choice([1, 2, 3, 7, 16, 30, 64, 100, 129, 200, 300, 700, 1500, 5000, 20000]))
        if D >= 5000: C = min(C, 3)
        if C == 8192 and D > 100: C = 70
        if tk == ol.T_DIAGGAUSS:
            tpar = 1.0 / np.logspace(0, rng.uniform(0, 2), D)
        q0 = rng.normal(size=(C, D)) / (np.sqrt(tpar) if tpar is not None else 1.0)
        H0 = float(rng.uniform(0.3, 1.2)) * max(D, 4) ** -0.25
    elif tk == ol.T_DENSEGAUSS:
        D = int(rng.choice([2, 5, 16, 17, 40, 64, 100, 128, 129, 160, 300]))
        if C == 8192 and D > 16: C = 70
        Q, _ = np.linalg.qr(rng.normal(size=(D, D))); lam = np.logspace(0, rng.uniform(0.5, 3), D)
        A = (Q * (1.0 / lam)) @ Q.T; A = 0.5 * (A + A.T); tpar = A.ravel()
        q0 = rng.normal(size=(C, D)) @ (Q * np.sqrt(lam)).T
        H0 = float(rng.uniform(0.3, 0.9))
    elif tk == ol.T_FUNNEL10:
        C = min(C, 70); D = 11; q0 = np.zeros((C, 11)); q0[:, 0] = 2.0 * rng.normal(size=C); q0[:, 1:] = np.exp(0.5 * q0[:, :1]) * rng.normal(size=(C, 10))
        H0 = float(rng.uniform(0.1, 0.5))
    elif tk == ol.T_STOCKWATSON:
        D = 3 * SWT; tpar = SWP; q0 = 0.1 * rng.normal(size=(C, D)); H0 = 0.002 if ig == 0 else 0.05; M = min(M, 4); C = min(C, 3); q0 = q0[:C]
    else:
        D = 2; q0 = rng.normal(size=(C, 2)); H0 = float(rng.uniform(0.2, 0.8))
    delta = float(rng.choice([0.3, 0.05, 0.01]))
    gen, gk = (ol.G_IDENTITY, 0) if rng.random() < 0.5 else ((ol.G_Q0_SUMSQ, 0) if rng.random() < 0.5 else (ol.G_FIRSTK, min(2, D)))
    try:
        e = wa.Engine(D, C, integrator=ig, H0=H0, delta0=delta, M=M, generated=gen, gen_k=gk, chain_offset=100 * it, warmupIter=warm, adaptQuantBits=QBITS if warm else 0)
        e.set_target(tk, tpar)
        s, d = e.run(q0, 777, numIter)
        s = s.transpose(1, 2).cpu().numpy(); d = d.cpu().numpy(); name = e.kernel_name; e.close()
    except NotImplementedError as ex:
        print(it, "unsupported:", tk, D, ig, ex); continue
    for k in sorted({0, C - 1}):
        o = ol.walnuts(tk, q0[k], integrator=ig, tpar=tpar, H0=H0, delta0=delta, numIter=numIter, M=M, seed=777,
                       chain=100 * it + k, generated=gen, gen_k=gk, warmupIter=warm, adaptQuantBits=QBITS if warm else 0)
        ok = np.array_equal(d[k][:, [6, 7, 19, 20]], o["diagnostics"][:, [6, 7, 19, 20]])
        with np.errstate(all="ignore"):
            a, b = s[k], o["samples"]
            # per-step state to rtol 1e-6, elements measured against the scale of their vector (an element that
            # happens to be near zero has no relative accuracy of its own); warm-up adaptation of H amplifies rounding
            # through the energy differences it is computed from, so those runs get 1e-4
            scale = np.nanmax(np.abs(b), axis=0, keepdims=True) if b.size else 1.0
            tol = (1e-4 if warm else 1e-6) if not QBITS else 1e-6  # (with the loop cut: 1e-6 with or without warm-up)
            close = ((np.isnan(a) & np.isnan(b)) | (np.abs(a - b) <= 1e-9 + tol * np.maximum(np.abs(b), scale))).all()
        if not (ok and close):
            bad += 1
            with np.errstate(all="ignore"):
                rel = np.nanmax(np.abs(a - b) / (1e-9 + np.abs(b)))
                relH = np.nanmax(np.abs(d[k][:, 15] - o["diagnostics"][:, 15]) / np.abs(o["diagnostics"][:, 15]))
                relE = np.nanmax(np.abs(d[k][:, 17] - o["diagnostics"][:, 17]) / (1e-300 + np.abs(o["diagnostics"][:, 17])))
            print(f"   max rel diff of draws {rel:.2e}, of H (col 15) {relH:.2e}, of the orbit energy error (col 17) {relE:.2e}")
            print(f"MISMATCH it={it} target={tk} D={D} ig={ig} C={C} M={M} iters={numIter} warm={warm} gen={gen} H0={H0:.4f} delta={delta} "
                  f"chain={k} kernel={name} steps_ok={ok} close={close}")
print(f"fuzz: {n} configurations, {bad} mismatching chains")
