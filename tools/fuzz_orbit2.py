"""Randomised differential test of the StockWatson kernels of round 5 -- wn_orbit2_kernel (trajectory wave + bookkeeping wave + helper waves,
wn_k_orbit2.hip) and wn_orbit_kernel's wide mode (WN_ORBIT_ONEWAVE=1) -- against the one-stream engine of the same layout
(WN_ORBIT_LEGACY=1): random series length T (ragged last lane, tSigma alone in its lane, ...), integrator, tree depth, step size from tame
to absurd (non-finite energies, refinement to maxC and beyond, passive ends), minC / maxC, delta, chains, generated kind.  The kernels
share the arithmetic of a micro step, of every wave sum and of every U-turn dot, so EVERY chain must agree bit for bit with the engine
(fixedLeapFrog: another instantiation of the engine, integers exactly and reals to 1e-7); a sample of chains is compared with the CPU
oracle as well (integers exactly, reals 1e-6) when the step size is tame.   python tools/fuzz_orbit2.py [n] [seed]   (GPU)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
import walnuts_amd as wa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
INT = [0, 1, 4, 5, 6, 7, 8, 9, 12, 13, 19, 20, 21, 22]
sw = np.load(os.path.join(ROOT, "tests", "golden", "stockwatson_data.npz"))
bad = 0; ndone = 0; nor = 0; t0 = time.time()
ONLY = set(int(x) for x in os.environ.get("WN_FUZZ_ONLY", "").split(",") if x)  # replay switch: run only these configurations of the stream


def run(env, T, C, tpar, q0, kw, numIter, off):
    # env: {"WN_<ROUTE>": "1"} -- the engine's route argument (wn_config.route), named as the environment switches of ABI <= 4 were
    e = wa.Engine(3 * T, C, chain_offset=off, route=[k[3:] for k in env], **kw)
    e.set_target(ol.T_STOCKWATSON, tpar)
    name = e.kernel_name
    s, d = e.run(q0, 777, numIter)
    s = s.cpu().numpy(); d = d.cpu().numpy(); e.close()
    return s, d, name


for it in range(n):
    T = int(rng.choice([252, 101, 30, 13, 64, 200, 255, 5]))
    ig = int(rng.choice([ol.I_FIXED, ol.I_ADAPT_D, ol.I_ADAPT_R2P], p=[.35, .3, .35]))
    C = int(rng.choice([1, 7, 40]))
    M = int(rng.integers(1, 12 if ig == ol.I_FIXED else 9))
    numIter = int(rng.integers(1, 7))
    minC = int(rng.choice([0, 0, 1, 3])); maxC = int(rng.choice([10, 6, 4])); maxC = max(maxC, minC)
    delta = float(rng.choice([0.3, 0.05, 1.0]))
    H0 = float(np.exp(rng.uniform(np.log(0.002), np.log(0.6)))) if rng.random() < 0.85 else float(rng.uniform(1.0, 20.0))
    gen, gk = (ol.G_IDENTITY, 0) if rng.random() < 0.5 else ((ol.G_Q0_SUMSQ, 0) if rng.random() < 0.5 else (ol.G_FIRSTK, 3))
    y = sw["y"][:T] if T <= len(sw["y"]) else np.concatenate([sw["y"], 0.5 * rng.normal(size=T - len(sw["y"]))])
    tpar = np.concatenate([[float(T)], y])
    q0 = 0.2 * rng.normal(size=(C, 3 * T)) if rng.random() < 0.7 else np.zeros((C, 3 * T))
    kw = dict(integrator=ig, H0=H0, delta0=delta, M=M, minC=minC, maxC=maxC, generated=gen, gen_k=gk)
    off = 1000 * it
    if ONLY and it not in ONLY:
        continue
    sl, dl, nl = run({"WN_ORBIT_LEGACY": "1"}, T, C, tpar, q0, kw, numIter, off)
    forms = [({}, "wn_orbit2_kernel"), ({"WN_ORBIT_ONEWAVE": "1", "WN_ORBIT_WIDE_ADAPT": "1"}, "wn_orbit_kernel")]
    for env, want in forms:
        s, d, name = run(env, T, C, tpar, q0, kw, numIter, off)
        assert want in name and "wn_wave_kernel" in nl, (name, nl)
        with np.errstate(all="ignore"):
            if ig == ol.I_FIXED:
                ok = np.array_equal(d[:, :, INT], dl[:, :, INT])
                rel = np.nanmax(np.abs(s - sl) / np.maximum(1e-3, np.abs(sl))) if s.size else 0.0
                ok = ok and (rel <= 1e-7 or np.isnan(rel) or H0 > 0.6)
            else:
                k17 = [c for c in range(24) if c != 17]
                ok = np.array_equal(d[:, :, k17], dl[:, :, k17], equal_nan=True) and np.array_equal(s, sl, equal_nan=True)
        if not ok:
            bad += 1
            cols = sorted(set(np.argwhere(~((d == dl) | (np.isnan(d) & np.isnan(dl))))[:, 2].tolist()))
            print(f"MISMATCH cfg {it} {want}: T {T} integ {ig} M {M} H0 {H0:.4g} delta {delta} minC {minC} maxC {maxC} gen {gen} iters {numIter} C {C}: diag columns {cols}", flush=True)
    ndone += 1
    if H0 <= 0.6 and M <= 7:  # the oracle on a couple of chains (its StockWatson chain costs what the CPU costs)
        for k in sorted(set([0, C - 1])):
            o = ol.walnuts(ol.T_STOCKWATSON, q0[k], integrator=ig, tpar=tpar, H0=H0, delta0=delta, numIter=numIter, M=M, minC=minC, maxC=maxC,
                           seed=777, chain=off + k, generated=gen, gen_k=gk)
            nor += 1
            with np.errstate(all="ignore"):
                close = lambda x, yy: np.all((np.isnan(x) & np.isnan(yy)) | (x == yy) | (np.abs(x - yy) <= 1e-9 + 1e-6 * np.abs(yy)))
                ok = np.array_equal(d[k][:, INT], o["diagnostics"][:, INT]) and bool(close(s[k].T, o["samples"])) and bool(close(d[k], o["diagnostics"]))
            if not ok:
                bad += 1
                print(f"ORACLE MISMATCH cfg {it} chain {k}: T {T} integ {ig} M {M} H0 {H0:.4g} delta {delta} minC {minC} maxC {maxC} gen {gen} iters {numIter}", flush=True)
                if ONLY:  # where they part: integers or reals, which iteration, by how much
                    od = o["diagnostics"]; os_ = o["samples"]
                    with np.errstate(all="ignore"):
                        rs = np.abs(s[k].T - os_) / np.maximum(1e-300, np.abs(os_)); rs = np.where(np.isnan(rs), 0.0, rs).max(axis=0 if os_.shape[1] == numIter else 1)
                        rd = np.abs(d[k] - od) / np.maximum(1e-300, np.abs(od)); rd = np.where(np.isnan(rd), 0.0, rd)
                    ib = int(np.argmax(np.any(d[k][:, INT] != od[:, INT], axis=1))) if not np.array_equal(d[k][:, INT], od[:, INT]) else -1
                    print(f"   draws: max rel diff per iteration {[float(f'{x:.2g}') for x in rs]}; first differing integer at iteration {ib}"
                          + (f" gpu {d[k][ib, INT]} oracle {od[ib, INT]}" if ib >= 0 else "") + f"; diagnostics columns over 1e-6: {sorted(set(np.argwhere(rd > 1e-6)[:, 1].tolist()))}"
                          f" max {rd.max():.3g}; stop codes {sorted(set(od[:, 19].astype(int).tolist()))}", flush=True)
                    if ib >= 0:
                        np.set_printoptions(linewidth=250, precision=6)
                        print("   gpu   ", d[k][ib], "\n   oracle", od[ib], "\n   |q| entering the iteration", float(np.abs(os_[:, ib - 1] if os_.shape[1] == numIter else os_[ib - 1]).max()) if ib > 0 else "q0", flush=True)
print(f"fuzz_orbit2: {ndone} configurations x 2 kernel forms against the one-stream engine, {nor} chains against the oracle, {bad} mismatches, {time.time() - t0:.0f} s")
