#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes + kernel trace of the default bench workload only: tools/prof_traffic_only.sh <tag>
set -u
TAG=${1:-r03}
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.log 2>&1
for pmc in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_$pmc -- python3 bench.py $ARGS > $OUT/pmc_$pmc.log 2>&1 || echo "pmc pass failed: $pmc"
done
WN_PROF_SIDE_SKIP=1 python3 tools/prof_traffic.py $TAG | head -20
