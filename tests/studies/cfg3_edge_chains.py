#!/usr/bin/env python3
"""BASELINE config 3 at its size (4096 chains, funnel10 / adaptLeapFrogR2P, M = 12, 20 iterations): which chains' integer diagnostics
differ between the trajectory-first kernel's two micro-step arithmetics (default: one row reduction per micro step; WN_ROUTE_ORBIT_EXACT:
the one-stream engine's two), what the oracle says about them, and how close to a rounding edge the deciding comparison sits
(Weak #2 / Next #3d: name the chains, show the edge).   python tests/studies/cfg3_edge_chains.py   (GPU box)"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as ol  # noqa: E402
import walnuts_amd as wa  # noqa: E402

SEED = 20251121
INT = [6, 7, 19, 20]


def funnel_q0(C, seed):
    rng = np.random.default_rng(seed)
    q0 = np.zeros((C, 11)); q0[:, 0] = 3.0 * rng.normal(size=C)
    q0[:, 1:] = np.exp(0.5 * q0[:, :1]) * rng.normal(size=(C, 10))
    return q0


def main():
    C, iters = 4096, 20
    for q0seed in (21, 22, 23):
        q0 = funnel_q0(C, q0seed)
        kw = dict(integrator=ol.I_ADAPT_R2P, H0=0.3, delta0=0.3, M=12, generated=ol.G_FIRSTK, gen_k=2)
        out = {}
        for nm, route in (("default", ["ORBIT_WIDE_ADAPT"]), ("exact", ["ORBIT_WIDE_ADAPT", "ORBIT_EXACT"])):
            e = wa.Engine(11, C, route=route, **kw).set_target(ol.T_FUNNEL10, None)
            s, d = e.run(q0, SEED, iters)
            out[nm] = (s.cpu().numpy(), d.cpu().numpy()); e.close()
        dd, de = out["default"][1], out["exact"][1]
        bad = np.flatnonzero((dd[..., INT] != de[..., INT]).any(axis=(1, 2)))
        print(f"q0 seed {q0seed}: chains whose integer columns differ between the two arithmetics: {bad.tolist()}")
        for c in bad:
            it = int(np.flatnonzero((dd[c][:, INT] != de[c][:, INT]).any(1))[0])
            o = ol.walnuts(ol.T_FUNNEL10, q0[c], integrator=ol.I_ADAPT_R2P, H0=0.3, delta0=0.3, numIter=iters, M=12, seed=SEED, chain=int(c),
                           generated=ol.G_FIRSTK, gen_k=2, trace_cap=400000)
            od = o["diagnostics"]
            who = "default" if (dd[c][:, INT] == od[:, INT]).all() else ("exact" if (de[c][:, INT] == od[:, INT]).all() else "neither")
            print(f"  chain {c}: first differing iteration {it + 1}: default {dd[c][it, INT]}, exact {de[c][it, INT]}, oracle {od[it, INT]} -> the oracle agrees with: {who}")
            # the deciding comparisons of an adaptive macro step are |Hams[0] - Hams[-1]| < delta (adaptiveIntegrators.py:86-90): the trace row holds
            # Ham0 (col 4) and, through lp (col 5) and v', the end energy; report the smallest margin of that iteration's calls
            tr = o["trace"][: o["ncalls"]]
            rows = tr[tr[:, 0] == it + 1]
            D = 11
            H1 = -rows[:, 5] + 0.5 * (rows[:, 12 + D:12 + 2 * D] ** 2).sum(1)
            marg = np.abs(np.abs(rows[:, 4] - H1) - 0.3) / 0.3
            k = int(np.argmin(marg))
            print(f"    iteration {it + 1}: {len(rows)} integrator calls; smallest relative margin of the accepted level's | |dH| - delta | / delta = {marg[k]:.3e} (call {k}: If={rows[k, 6]:.0f} Ib={rows[k, 7]:.0f} c={rows[k, 8]:.0f})")


if __name__ == "__main__":
    main()
