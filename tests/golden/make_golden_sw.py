"""Fixture for the StockWatson target (BASELINE.json config 5): the observation vector y[T] of the
reference's example data file, stored as data.  Run in the build container only:
    python3 -B tests/golden/make_golden_sw.py
"""
import json
import os

import numpy as np

SRC = "/root/reference/WALNUTSpy_examples/StockWatson/swdata.json"
d = json.load(open(SRC))
y = np.asarray(d["y"], dtype=np.float64)
assert y.size == int(d["T"])
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stockwatson_data.npz")
np.savez(out, T=np.array([int(d["T"])]), y=y)
print("wrote", out, y.size)
