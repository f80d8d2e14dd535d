#!/bin/bash
mkdir -p gpurun_out
WN_PROF_SKIP_DEFAULT=1 WN_PROF_SIDE=stockwatson WN_PROF_BENCH="default stockwatson" bash tools/prof_round.sh r05b > gpurun_out/prof_round_r05b.log 2>&1
for r in fixed R2P; do
  O=gpurun_out/prof_r05b_sw_$r; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --workload stockwatson --sw-run $r --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_trace.log 2>&1
  python3 tools/prof_summary.py $O > $O/summary.txt 2>&1
done
bash tools/sw_stamps.sh > gpurun_out/r05b_orbit2_stamps.txt 2>&1
tail -3 gpurun_out/prof_round_r05b.log
