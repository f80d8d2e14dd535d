#!/bin/bash
# Counter passes of the pair kernel on the bench workload: tools/fast4_pmc2.sh <out> <variant|base> "<counters pass 1>" "<counters pass 2>" ...
OUT=$1; V=$2; shift 2
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p $(dirname $OUT); : > $OUT
if [ "$V" = base ]; then unset WN_LIB; else export WN_LIB=$V; fi
n=0
for pass in "$@"; do
  n=$((n+1)); D=/tmp/pmc2_$n; rm -rf $D
  rocprofv3 --pmc $pass --output-format csv -d $D -- python3 bench.py $WN_BENCH_ARGS --no-extras --no-cpu-baseline --steps 1 --warmup 0 > /tmp/pmc2_$n.log 2>&1 || { echo "pass '$pass' failed: $(tail -2 /tmp/pmc2_$n.log | tr '\n' ' ')" >> $OUT; continue; }
  python3 - $D >> $OUT <<'PY'
import csv, glob, sys, os
acc = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "wn_fast" not in r["Kernel_Name"]: continue
        k = (r["Dispatch_Id"], r["Counter_Name"])
        acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"])
by = {}
for (d, c), v in acc.items(): by.setdefault(c, []).append(v)
for c, v in sorted(by.items()): print(f"{c:28s} n={len(v)} mean={sum(v)/len(v):.6g}")
PY
done
cat $OUT
