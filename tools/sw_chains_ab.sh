#!/bin/bash
# StockWatson at more chains than CUs: the multi-wave-per-chain kernel (wn_k_orbit2.hip) against the one-stream engine (--route ORBIT_LEGACY)
for C in 256 512 1024 2048; do for r in D fixed; do for leg in 0 1; do
  if [ $leg = 1 ]; then RT="--route ORBIT_LEGACY"; else RT=""; fi
  timeout 300 python bench.py $RT --workload stockwatson --sw-run $r --chains $C --iters 8 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('C=$C $r', d['config']['kernel'], '%.3e'%d['value'], 'ms', '%.1f'%d['ms_per_step'])
"; done; done; done
