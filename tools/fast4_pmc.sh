#!/bin/bash
# Instruction counts of the pair kernel per launch: [WN_BENCH_ARGS="--chains 1024"] tools/fast4_pmc.sh <out> <variants...>  (base = the shipped library)
OUT=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p $(dirname $OUT); : > $OUT
for v in "$@"; do
  if [ "$v" = base ]; then unset WN_LIB; else export WN_LIB=$v; fi
  D=/tmp/pmc_$v; rm -rf $D
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $D -- python3 bench.py $WN_BENCH_ARGS --no-extras --no-cpu-baseline --steps 1 --warmup 0 > /tmp/pmc_$v.log 2>&1
  echo "== variant $v" >> $OUT
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
for c, v in sorted(by.items()): print(f"{c:20s} n={len(v)} mean={sum(v)/len(v):.6g}")
PY
done
cat $OUT
