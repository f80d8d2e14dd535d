#!/bin/bash
# A/B of the pair kernel's diagnostic variants on the bench workload (run on the GPU box): tools/fast4_ab.sh <out> <variants...>
OUT=$1; shift
mkdir -p $(dirname $OUT)
: > $OUT
for v in "$@"; do
  if [ "$v" = base ]; then unset WN_LIB; else export WN_LIB=$v; fi
  echo "== variant $v" >> $OUT
  timeout 300 python3 bench.py --no-extras --no-cpu-baseline --steps 2 --warmup 1 2>&1 | grep -E "WN_STAMPS|\"value\"" | \
    python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('value %.4g steps/s  kernel_ms %.2f  frac %.3f  %s' % (j['value'], j['kernel_ms_avg'], j['roofline']['frac'], j['config']['kernel']))
    else: print(l.rstrip()[:300])
" >> $OUT
done
cat $OUT
