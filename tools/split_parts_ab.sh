# dense1000 (BASELINE config 4) cut into sub-batches on streams and host threads of their own: tools/split_parts_ab.sh [trace]
run() { echo "WN_SPLIT_PARTS=${WN_SPLIT_PARTS:-1} WN_SPLIT_KS=${WN_SPLIT_KS:-1}"; python3 bench.py --workload dense1000 --no-extras --no-cpu-baseline --steps 3 2>/dev/null | python3 -c 'import sys,json; j=json.loads(sys.stdin.readline()); print("  %.4g steps/s, %.1f ms per launch, draws %s" % (j["value"], j["ms_per_step"], j.get("draws_sha256","")[:12]))'; }
run
WN_SPLIT_PARTS=2 WN_SPLIT_KS=1 run
WN_SPLIT_PARTS=2 WN_SPLIT_KS=2 run
WN_SPLIT_PARTS=4 WN_SPLIT_KS=1 run
if [ "$1" = trace ]; then
  R=$PWD; cd /tmp; export TMPDIR=/tmp
  for ks in 1 2; do
    export WN_SPLIT_PARTS=2 WN_SPLIT_KS=$ks
    rm -rf /tmp/trp$ks
    rocprofv3 --kernel-trace --output-format csv -d /tmp/trp$ks -- python3 $R/bench.py --workload dense1000 --no-extras --no-cpu-baseline --steps 1 --warmup 0 --iters 10 > /dev/null 2>&1
    echo "== kernel trace, WN_SPLIT_PARTS=2 WN_SPLIT_KS=$ks (10 iterations) =="; python3 $R/tools/trace_parts.py /tmp/trp$ks
  done
fi
