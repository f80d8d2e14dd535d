#!/usr/bin/env python3
"""Golden draws of the pseudocode-faithful variant walnuts() (walnuts/walnuts.py:362), generated
by running the UNMODIFIED reference in this container with a seeded numpy Generator.

Run:  PYTHONDONTWRITEBYTECODE=1 python3 -B tests/golden/make_golden_walnuts.py

BASELINE configs[0]: 10-D standard normal, 4 chains (seeds 123..126) x 1000 draws with
test/test.py's parameters (macro_step=2.0, max_nuts_depth=10, max_error=0.1).  The reference
module is loaded by path because `import walnuts` pulls bridgestan.
"""
import importlib.util
import os
import sys
import time

sys.dont_write_bytecode = True
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_walnuts", "/root/reference/walnuts/walnuts.py")
ref = importlib.util.module_from_spec(spec)
import tqdm  # noqa: F401
spec.loader.exec_module(ref)
ref.trange = lambda n, **k: range(n)  # no progress bar in the fixture run
sys.path.insert(0, "/root/reference/test")
import targets as tt  # noqa: E402

out = {}
t0 = time.time()
for seed in (123, 124, 125, 126):
    rng = np.random.default_rng(seed)
    d = ref.walnuts(rng, np.zeros(10), tt.standard_normal_lpdf, tt.standard_normal_grad, np.ones(10),
                    2.0, 10, 0.1, 0, 1000)
    out[f"std10_seed{seed}"] = d
    out[f"std10_seed{seed}_next_uniform"] = np.array([rng.uniform()])  # stream position after the run
print("config-1 runs: %.1f s" % (time.time() - t0))
rng = np.random.default_rng(7)
out["funnel3_seed7"] = ref.walnuts(rng, np.array([0.5, 0.1, -0.2]), tt.funnel_lpdf, tt.funnel_grad,
                                   np.array([1.0, 2.0, 0.5]), 1.5, 8, 0.2, 20, 150)
out["funnel3_seed7_next_uniform"] = np.array([rng.uniform()])
rng = np.random.default_rng(8)
out["corr2_seed8"] = ref.walnuts(rng, np.array([1.0, 0.0]), tt.correlated_normal_lpdf,
                                 tt.correlated_normal_grad, np.ones(2), 1.8, 8, 0.1, 0, 200)
out["corr2_seed8_next_uniform"] = np.array([rng.uniform()])
rng = np.random.default_rng(9)
out["rosen2_seed9"] = ref.walnuts(rng, np.array([0.3, -0.2]), tt.rosenbrock_lpdf, tt.rosenbrock_grad,
                                  np.ones(2), 1.0, 8, 0.1, 0, 200)
out["rosen2_seed9_next_uniform"] = np.array([rng.uniform()])
np.savez_compressed(os.path.join(HERE, "walnuts_draws.npz"), **out)
for k, v in out.items():
    if not k.endswith("uniform"):
        print(k, v.shape, "mean", np.round(v.mean(0)[:3], 3), "std", np.round(v.std(0)[:3], 3))
