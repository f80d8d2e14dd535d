#!/bin/bash
# usage: ab_c.sh out chains variants...
OUT=$1; C=$2; shift; shift
: > $OUT
for v in "$@"; do
  if [ "$v" = base ]; then unset WN_LIB; else export WN_LIB=$v; fi
  echo "== variant $v chains $C" >> $OUT
  timeout 300 python3 bench.py --no-extras --no-cpu-baseline --steps 2 --warmup 1 --chains $C 2>&1 | grep "\"value\"" | python3 -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('value %.4g steps/s  kernel_ms %.2f' % (j['value'], j['kernel_ms_avg']))
" >> $OUT
done
cat $OUT
