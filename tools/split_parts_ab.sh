#!/bin/bash
# dense1000 (BASELINE config 4) cut into sub-batches on streams and host threads of their own: tools/split_parts_ab.sh [trace]
run() { echo "split_parts=${1:-1} split_ks=${2:-1}"; python3 bench.py --split-parts ${1:-0} --split-ks ${2:-0} --workload dense1000 --no-extras --no-cpu-baseline --steps 3 2>/dev/null | python3 -c 'import sys,json; j=json.loads(sys.stdin.readline()); print("  %.4g steps/s, %.1f ms per launch, draws %s" % (j["value"], j["ms_per_step"], j.get("draws_sha256","")[:12]))'; }
TRACE=$1
run
run 2 1
run 2 2
run 4 1
if [ "$TRACE" = trace ]; then
  R=$PWD; cd /tmp; export TMPDIR=/tmp
  for ks in 1 2; do
    rm -rf /tmp/trp$ks
    rocprofv3 --kernel-trace --output-format csv -d /tmp/trp$ks -- python3 $R/bench.py --split-parts 2 --split-ks $ks --workload dense1000 --no-extras --no-cpu-baseline --steps 1 --warmup 0 --iters 10 > /dev/null 2>&1
    echo "== kernel trace, split_parts=2 split_ks=$ks (10 iterations) =="; python3 $R/tools/trace_parts.py /tmp/trp$ks
  done
fi
