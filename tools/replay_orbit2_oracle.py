"""CPU-only replay of one configuration of tools/fuzz_orbit2.py's stream (seed 31, configuration 739, chain 0) on the oracle with integrator-call traces:
what the oracle saw where it and the GPU kernels parted (profiles/r05_fuzz_final.txt).   python tools/replay_orbit2_oracle.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/tests")
import numpy as np, oracle_lib as ol
rng = np.random.default_rng(31)
sw = np.load(ROOT+"/tests/golden/stockwatson_data.npz")
for it in range(740):
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
print(it, T, ig, C, M, numIter, minC, maxC, delta, H0, gen)
o = ol.walnuts(ol.T_STOCKWATSON, q0[0], integrator=ig, tpar=tpar, H0=H0, delta0=delta, numIter=numIter, M=M, minC=minC, maxC=maxC, seed=777, chain=1000*it, generated=gen, gen_k=gk, trace_cap=200)
tr = o["trace"]; d = 3*T
np.set_printoptions(linewidth=220, precision=6)
for r in tr:
    q1 = r[12:12+d]; v1 = r[12+d:12+2*d]
    print("iter %d doubl %d xi %d h %.4g Ham0 %.6g lp %.6g If %d Ib %d c %d lwt %.3g nF %d nB %d |q1| %.3g |v1| %.3g finite q %s v %s  H1 %.6g" % (r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[10], r[11], np.nanmax(np.abs(q1)), np.nanmax(np.abs(v1)), np.isfinite(q1).all(), np.isfinite(v1).all(), -r[5] + 0.5*np.dot(v1, v1)))
print(o["diagnostics"][:, [19, 6, 7]])
