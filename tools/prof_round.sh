#!/bin/bash
# One round's measurement set (run on the GPU box through gpurun): tools/prof_round.sh <tag>
#  - tools/prof_bench.sh <tag> on the default workload (kernel trace + every PMC pass) -> gpurun_out/prof_<tag>/summary.txt
#  - kernel trace + FETCH_SIZE / WRITE_SIZE passes of the side workloads (funnel, dense1000, sw)
#  - gpurun_out/<tag>_traffic.json, <tag>_traffic_side.json, and the bench lines <tag>_bench_*.json
# Copy what is to be judged into profiles/.
set -u
TAG=${1:-r02}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
[ -n "${WN_PROF_SKIP_DEFAULT:-}" ] || bash tools/prof_bench.sh $TAG --no-extras > gpurun_out/prof_$TAG.log 2>&1
for wl in ${WN_PROF_SIDE:-funnel dense1000 stockwatson}; do
  O=gpurun_out/prof_${TAG}_$wl; mkdir -p $O
  ARGS="--workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py $ARGS > $O/bench_trace.log 2>&1
  for pmc in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    n=$(echo $pmc | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $pmc --output-format csv -d $O/pmc_$n -- python3 bench.py $ARGS > $O/pmc_$n.log 2>&1 || echo "pmc pass failed: $wl $pmc"
  done
  python3 tools/prof_summary.py $O > $O/summary.txt 2>&1
done
python3 tools/prof_traffic.py $TAG
for wl in ${WN_PROF_BENCH:-default funnel dense1000 stockwatson}; do
  a="--workload $wl"; [ $wl = default ] && a=""
  python3 bench.py $a > gpurun_out/${TAG}_bench_$wl.json 2> gpurun_out/${TAG}_bench_$wl.err
  tail -c 600 gpurun_out/${TAG}_bench_$wl.json
done
