# dense1000 (BASELINE config 4) cut into sub-batches on streams of their own: tools/split_parts_ab.sh
run() { echo "WN_SPLIT_PARTS=${WN_SPLIT_PARTS:-default} WN_SPLIT_KS=${WN_SPLIT_KS:-default}"; python bench.py --workload dense1000 --no-extras --no-cpu-baseline --steps 3 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.readline()); print(j["value"], j["ms_per_step"], j.get("draws_sha256","")[:12])'; }
WN_SPLIT_PARTS=1 run
WN_SPLIT_PARTS=2 WN_SPLIT_KS=1 run
run
