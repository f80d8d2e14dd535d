"""Randomised differential test of the adaptive sampling runs of Gaussian targets on the trajectory-first kernel (round 5: std / diagonal
Gaussians up to 512 dimensions, dense ones up to 128: one register per lane up to 16 dimensions, wide mode beyond, dense: matrix in LDS;
with and without warm-up iterations) against the kernels these runs used before (WN_DIAG_COOP=1 / WN_DENSE_COOP=1: cooperative MFMA layout,
one-stream engine of the wave layout) -- integer diagnostics exactly, reals to 1e-6 -- and against the CPU oracle on two chains.
python tools/fuzz_gauss_orbit.py [n] [seed]   (GPU)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
import walnuts_amd as wa
from bench import make_dense_problem

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
QBITS = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # adapt_quant_bits of the warm-up runs, engine and oracle alike (round 6: 16 cuts the feedback loop through H / delta; 0 = the reference's arithmetic)
INT = [0, 1, 4, 5, 6, 7, 8, 9, 12, 13, 19, 20, 21, 22]
bad = 0; badl = 0; done = 0; nor = 0; nwarm = 0; kernels = {}; t0 = time.time()


def run(env, D, C, target, tpar, q0, kw, iters, off):
    # env: {"WN_<ROUTE>": "1"} -- the engine's route argument (wn_config.route), named as the environment switches of ABI <= 4 were
    e = wa.Engine(D, C, chain_offset=off, route=[k[3:] for k in env], **kw)
    e.set_target(target, tpar)
    name = e.kernel_name
    s, d = e.run(q0, 4711, iters)
    s = s.cpu().numpy(); d = d.cpu().numpy(); e.close()
    return s, d, name


for it in range(n):
    kind = int(rng.choice([ol.T_STDGAUSS, ol.T_DIAGGAUSS, ol.T_DENSEGAUSS], p=[.35, .3, .35]))
    if kind == ol.T_DENSEGAUSS:
        D = int(rng.choice([2, 3, 8, 16, 17, 40, 64, 65, 100, 128]))
    else:
        D = int(rng.choice([2, 5, 10, 16, 17, 33, 64, 100, 128, 129, 200, 256, 257, 400, 512]))
    ig = int(rng.choice([ol.I_ADAPT_D, ol.I_ADAPT_R2P]))
    C = int(rng.choice([1, 7, 33]))
    M = int(rng.integers(1, 9))
    iters = int(rng.integers(2, 14))
    warm = int(rng.choice([0, 0, iters // 2, iters]))
    minC = int(rng.choice([0, 0, 1])); maxC = int(rng.choice([10, 4])); delta = float(rng.choice([0.3, 0.05, 1.0]))
    H0 = float(np.exp(rng.uniform(np.log(0.05), np.log(1.2)))) * (D ** -0.25 if kind != ol.T_DENSEGAUSS else 1.0)
    gen, gk = (ol.G_IDENTITY, 0) if rng.random() < 0.5 else ((ol.G_Q0_SUMSQ, 0) if rng.random() < 0.5 else (ol.G_FIRSTK, min(2, D)))
    if kind == ol.T_DENSEGAUSS:
        A, q0 = make_dense_problem(D, float(rng.choice([10.0, 1e3])), C, int(rng.integers(1 << 30)))
        tpar = A
    elif kind == ol.T_DIAGGAUSS:
        tpar = np.geomspace(0.2, float(rng.choice([2.0, 20.0])), D)
        q0 = rng.normal(size=(C, D)) / np.sqrt(tpar)
    else:
        tpar = None
        q0 = rng.normal(size=(C, D))
    kw = dict(integrator=ig, H0=H0, delta0=delta, M=M, minC=minC, maxC=maxC, generated=gen, gen_k=gk, warmupIter=warm, adaptQuantBits=QBITS if warm else 0)
    off = 1000 * it
    s, d, name = run({}, D, C, kind, tpar, q0, kw, iters, off)
    sl, dl, namel = run({"WN_DIAG_COOP": "1", "WN_DENSE_COOP": "1"}, D, C, kind, tpar, q0, kw, iters, off)
    kernels[name.split(",adapt")[0]] = kernels.get(name.split(",adapt")[0], 0) + 1
    if "wn_orbit_kernel" not in name or "wn_orbit_kernel" in namel:
        print(f"cfg {it}: kernels {name} / {namel} (not the pair this fuzzer is about)", flush=True)
        continue
    done += 1; nwarm += int(warm > 0)
    with np.errstate(all="ignore"):
        close = lambda x, y: bool(np.all((np.isnan(x) & np.isnan(y)) | (x == y) | (np.abs(x - y) <= 1e-9 + 1e-6 * np.abs(y))))
        ok = np.array_equal(d[:, :, INT], dl[:, :, INT]) and close(s, sl) and close(d, dl)
    if not ok:
        bad += 1
        print(f"MISMATCH cfg {it}: {name} vs {namel}: target {kind} D {D} integ {ig} M {M} H0 {H0:.4g} delta {delta} minC {minC} maxC {maxC} gen {gen} iters {iters} warm {warm} C {C}: "
              f"ints equal {np.array_equal(d[:, :, INT], dl[:, :, INT])}", flush=True)
    for k in sorted(set([0, C - 1])):
        o = ol.walnuts(kind, q0[k], integrator=ig, tpar=None if tpar is None else np.asarray(tpar).ravel(), H0=H0, delta0=delta, numIter=iters, warmupIter=warm, M=M,
                       minC=minC, maxC=maxC, seed=4711, chain=off + k, generated=gen, gen_k=gk, adaptQuantBits=QBITS if warm else 0)
        nor += 1
        with np.errstate(all="ignore"):
            ok = np.array_equal(d[k][:, INT], o["diagnostics"][:, INT]) and close(s[k].T, o["samples"]) and close(d[k], o["diagnostics"])
            okl = np.array_equal(dl[k][:, INT], o["diagnostics"][:, INT]) and close(sl[k].T, o["samples"]) and close(dl[k], o["diagnostics"])
        if not okl:
            badl += 1  # (the kernel these runs used before, against the oracle: the rate a warm-up run's feedback makes of last-bit differences)
        if not ok:
            bad += 1
            print(f"ORACLE MISMATCH cfg {it} chain {k}: {name}: target {kind} D {D} integ {ig} M {M} H0 {H0:.4g} delta {delta} iters {iters} warm {warm}"
                  f" (previous kernel against the oracle: {'ok' if okl else 'MISMATCH too'})", flush=True)
print(f"fuzz_gauss_orbit (adapt_quant_bits {QBITS}): {done} configurations ({nwarm} with warm-up iterations) against the previous kernels, {nor} chains against the oracle, {bad} mismatches "
      f"(the previous kernels against the oracle: {badl} chains), {time.time() - t0:.0f} s; kernels: {kernels}")
