#!/usr/bin/env python3
"""Warm-up tolerance contract, measured (Next #3a): the engine (GPU, through the C ABI) against the oracle on the same
random streams, WITH warm-up adaptation of H and delta (WALNUTS.py:120-125, 701-712), chain by chain: the fraction of chains whose adapted
H / delta, draws and integer diagnostics still agree after k iterations -- beside the SAME curve for the oracle against itself with one
input coordinate moved by one ulp (tests/studies/warmup_sensitivity.py), i.e. beside what the reference's own arithmetic does to a last-bit
difference -- and both again with the opt-in `adapt_quant_bits` (ABI 5; adapted H / delta rounded to b mantissa bits on both sides).

  python tests/studies/warmup_curve_gpu.py [chains] [iterations] [bits]      (GPU box; the oracle runs on the host cores)
"""
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as ol  # noqa: E402
from warmup_sensitivity import SEED, _one, curve, problems  # noqa: E402


def oracle_runs(pool, p, q0, C, iters, warm, bits):
    kw = dict(p["kw"], adaptQuantBits=bits)
    r = pool.map(_one, [(p["target"], p["tpar"], q0[c], c, iters, warm, kw) for c in range(C)], chunksize=4)
    return np.stack([x[0] for x in r]), np.stack([x[1] for x in r])


def main():
    C = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    bits_list = [0] + [int(b) for b in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["20"])]
    marks = [k for k in (1, 2, 5, 10, 12, 15, 20, 30, 50, 100, 150, 200, 500, 1000) if k <= iters]
    probs = problems(C)
    pool = Pool(min(os.cpu_count(), 128))  # forked before any GPU work
    import walnuts_amd as wa
    for name, p in probs.items():
        for bits in bits_list:
            t0 = time.time()
            e = wa.Engine(p["q0"].shape[1], C, warmupIter=iters, adaptQuantBits=bits, **p["kw"]).set_target(p["target"], p["tpar"])
            s, d = e.run(p["q0"], SEED, iters)
            sg = s.cpu().numpy(); dg = d.cpu().numpy(); kname = e.kernel_name; e.close()
            do, so = oracle_runs(pool, p, p["q0"], C, iters, iters, bits)
            qb = p["q0"].copy(); qb[:, 0] = np.nextafter(qb[:, 0], np.inf)
            dp, sp = oracle_runs(pool, p, qb, C, iters, iters, bits)
            cg, cp = curve(dg, sg, do, so), curve(dp, sp, do, so)
            print(f"{name}: {C} chains x {iters} warm-up iterations, adapt_quant_bits = {bits}; kernel {kname}  ({time.time() - t0:.0f} s)")
            print("  after iteration                      " + " ".join(f"{k:>6d}" for k in marks))
            for key, lab in (("H6", "H, delta within 1e-6"), ("H4", "H, delta within 1e-4"), ("S6", "draws within 1e-6   "), ("S4", "draws within 1e-4   "), ("int", "integer cols equal  ")):
                print(f"  {lab}  GPU vs oracle    " + " ".join(f"{cg[key][k - 1]:6.3f}" for k in marks))
                print(f"  {lab}  oracle vs +1 ulp " + " ".join(f"{cp[key][k - 1]:6.3f}" for k in marks))
            # is the first divergence a discrete event (a comparison that went the other way) or continuous growth?
            rel = np.abs(dg[..., 15] - do[..., 15]) / np.abs(do[..., 15])
            med = [float(np.median(rel[:, k - 1])) for k in marks]
            p90 = [float(np.quantile(rel[:, k - 1], 0.9)) for k in marks]
            print("  |H_gpu - H_oracle| / H, median       " + " ".join(f"{x:6.0e}" for x in med))
            print("  |H_gpu - H_oracle| / H, 90 %         " + " ".join(f"{x:6.0e}" for x in p90))
            sys.stdout.flush()


if __name__ == "__main__":
    main()
