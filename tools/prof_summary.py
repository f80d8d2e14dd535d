#!/usr/bin/env python3
"""Summarise a tools/prof_bench.sh output directory: per-kernel time from the kernel trace and
per-dispatch PMC counters of the sampler kernel (wn_*_kernel)."""
import csv
import glob
import os
import sys

out = sys.argv[1]


def rows(pattern):
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(f) as fh:
            yield from csv.DictReader(fh)


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            print(f"{r.get('Name','?')[:70]:70s} calls={r.get('Calls')} total_ns={r.get('TotalDurationNs')} "
                  f"avg_ns={r.get('AverageNs')} pct={r.get('Percentage')}")
print("== sampler-kernel dispatches (kernel trace), grouped by kernel ==")
grp = {}
for r in rows("trace/**/*kernel_trace.csv"):
    if "wn_" in r.get("Kernel_Name", ""):
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        g = grp.setdefault(r["Kernel_Name"][:70], dict(n=0, tot=0, r=r))
        g["n"] += 1; g["tot"] += dur
for k, g in grp.items():
    r = g["r"]
    print(f"{k} calls={g['n']} avg_us={g['tot']/g['n']/1e3:.2f} total_ms={g['tot']/1e6:.3f} vgpr={r.get('VGPR_Count')} "
          f"accum={r.get('Accum_VGPR_Count')} sgpr={r.get('SGPR_Count')} lds={r.get('LDS_Block_Size')} "
          f"scratch={r.get('Scratch_Size')} wg={r.get('Workgroup_Size')}")
print("== code-object metadata (tools/kernel_meta.sh; the trace's VGPR_Count column above is NOT the allocation: cite these) ==")
import subprocess
for unit in ("wn_k_fast", "wn_k_fast4", "wn_k_wave", "wn_k_orbit", "wn_k_orbit2", "wn_k_split", "wn_k_gemm", "wn_k_coop"):
    used = [k for k in grp if unit.replace("wn_k_", "wn_") + "_kernel" in k or (unit == "wn_k_gemm" and "dense_grad" in k)]
    if not used:
        continue
    try:
        meta = subprocess.run([os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernel_meta.sh"), unit], capture_output=True, text=True, timeout=900).stdout
        print(f"-- {unit}.hip")
        print(meta.rstrip())
    except Exception as ex:
        print(f"-- {unit}.hip: {ex!r}")
print("== PMC counters per dispatch, grouped by kernel (mean over dispatches) ==")
acc = {}
names = {}
for r in rows("pmc_*/**/*counter_collection.csv"):
    if "wn_" not in r.get("Kernel_Name", ""):
        continue
    key = (r["Kernel_Name"][:50], r["Dispatch_Id"], r["Counter_Name"])
    acc[key] = acc.get(key, 0.0) + float(r["Counter_Value"])
byname = {}
for (k, d, n), v in acc.items():
    byname.setdefault((k, n), []).append(v)
for (k, n) in sorted(byname):
    vals = byname[(k, n)]
    print(f"{k:50s} {n:28s} n={len(vals)} mean={sum(vals)/len(vals):.6g}")
