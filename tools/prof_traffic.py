#!/usr/bin/env python3
"""HBM traffic records from the FETCH_SIZE / WRITE_SIZE passes of tools/prof_round.sh:
gpurun_out/<tag>_traffic.json (default workload, bytes per launch) and <tag>_traffic_side.json (bytes per unit of
the side workloads).  rocprofv3 reports KB; bytes = value * 1024 (MI355X_MICROARCH.md, HBM section)."""
import csv
import glob
import json
import os
import re
import sys

tag = sys.argv[1]
root = os.environ.get("GRAFT_REPO_ROOT", ".")


def counters(d):
    """{kernel: {counter: mean per dispatch}} and {kernel: (calls, avg_ms)} for the wn_* kernels of one output dir"""
    acc = {}
    for f in glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "wn_" not in k:
                continue
            key = (k.split("(")[0][:60], r["Counter_Name"], r["Dispatch_Id"])
            acc[key] = acc.get(key, 0.0) + float(r["Counter_Value"])
    by = {}
    for (k, c, _), v in acc.items():
        by.setdefault(k, {}).setdefault(c, []).append(v)
    out = {k: {c: (sum(v) / len(v), len(v)) for c, v in cs.items()} for k, cs in by.items()}
    tr = {}
    for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "wn_" in k:
                t = tr.setdefault(k.split("(")[0][:60], [0, 0.0])
                t[0] += 1; t[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    return out, {k: (n, tot / n) for k, (n, tot) in tr.items()}


def bench_line(d):
    for line in open(os.path.join(d, "bench_trace.log")):
        if line.startswith("{"):
            return json.loads(line)
    return None


go = os.path.join(root, "gpurun_out")
c, tr = counters(os.path.join(go, f"prof_{tag}"))
bl = None
try:
    bl = bench_line(os.path.join(go, f"prof_{tag}"))
except Exception:
    pass
for k, v in c.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        if bl and "wn_fast" not in k and "wn_coop" not in k:
            continue  # (the headline workload's own kernel, not torch's or the fill kernel)
        rec = {"comment": "HBM traffic of the default bench workload's kernel: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate "
                          "passes (tools/prof_round.sh), KB * 1024, mean per dispatch.  The guide's gfx950 correction (FETCH_SIZE counts half of wide "
                          "16-B/lane streaming reads) is not applied: this kernel's global accesses are 8 B/lane.",
               "kernel": k,
               # what bench.py matches on: the workload of the traced pass (tools/prof_bench.sh) and the engine's kernel name
               "workload": None if not bl else {"dim": bl["config"]["dim"], "chains": bl["config"]["chains_per_gpu"], "iters": bl["config"]["iters_per_step"],
                                                "kernel": bl["config"]["kernel"]},
               "FETCH_SIZE_KB": v["FETCH_SIZE"][0], "WRITE_SIZE_KB": v["WRITE_SIZE"][0],
               "bytes_per_launch": (v["FETCH_SIZE"][0] + v["WRITE_SIZE"][0]) * 1024.0,
               "kernel_avg_ms_rocprof": tr.get(k, (0, None))[1]}
        json.dump(rec, open(os.path.join(go, f"{tag}_traffic.json"), "w"), indent=1)
        print(json.dumps(rec))
side = {"comment": "HBM bytes per unit of work of the side workloads: FETCH_SIZE + WRITE_SIZE (KB * 1024) summed over the sampler kernels of one "
                   "timed launch, divided by the units the bench line of the same run reports (tools/prof_round.sh)."}
for wl, unit in (("funnel", "bytes_per_gradient_evaluation"), ("dense1000", "bytes_per_leapfrog_step"), ("stockwatson", "bytes_per_gradient_evaluation")):
    d = os.path.join(go, f"prof_{tag}_{wl}")
    if not os.path.isdir(d):
        continue
    c, tr = counters(d)
    b = bench_line(d)
    if not b:
        continue
    units = b["value"] * b["ms_per_step"] * 1e-3  # units of one timed step (= one launch per rank)
    tot = 0.0; src = []
    for k, v in c.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            calls = tr.get(k, (0, 0))[0]
            nlaunch = b["steps"] + b["warmup"]
            per_launch = calls / nlaunch if calls else 1.0  # dispatches of this kernel per bench step
            # gfx950: FETCH_SIZE reports exactly half of the bytes of wide coalesced streaming reads (16 B per lane; MI355X_MICROARCH.md, HBM
            # section).  wn_dense_grad_kernel reads both operands with 16-byte fragment loads (wn_k_gemm.hip WN_GEMM_LOAD): doubled.  The other
            # sampler kernels' global accesses are 8 B per lane: as reported.
            fcorr = 2.0 if "wn_dense_grad_kernel" in k else 1.0
            kb = (fcorr * v["FETCH_SIZE"][0] + v["WRITE_SIZE"][0]) * per_launch
            tot += kb * 1024.0
            src.append(f"{k}: {fcorr * v['FETCH_SIZE'][0]:.1f} KB read{' (FETCH_SIZE x 2: 16-B/lane loads)' if fcorr > 1 else ''} + {v['WRITE_SIZE'][0]:.1f} KB written per dispatch x {per_launch:.1f} dispatches")
    side[wl] = {unit: tot / units, "units_per_launch": units, "source": "; ".join(src)}
json.dump(side, open(os.path.join(go, f"{tag}_traffic_side.json"), "w"), indent=1)
print(json.dumps(side, indent=1))
