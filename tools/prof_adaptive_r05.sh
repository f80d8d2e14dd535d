#!/bin/bash
# rocprofv3 kernel traces of the adaptive runs that moved to the trajectory-first kernel in round 5 (run on the GPU box through gpurun)
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r05_adapt_std gpurun_out/prof_r05_adapt_dense gpurun_out/prof_r05_warm
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05_adapt_std/trace -- python3 tools/adapt_std.py 4096 10,100,256 > gpurun_out/prof_r05_adapt_std/bench_trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05_adapt_dense/trace -- python3 tools/adapt_dense100.py > gpurun_out/prof_r05_adapt_dense/bench_trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05_warm/trace -- python3 tools/warm_orbit_ab.py > gpurun_out/prof_r05_warm/bench_trace.log 2>&1
for d in prof_r05_adapt_std prof_r05_adapt_dense prof_r05_warm; do python3 tools/prof_summary.py gpurun_out/$d > gpurun_out/$d/summary.txt 2>&1; head -12 gpurun_out/$d/summary.txt | cut -c1-160; done
