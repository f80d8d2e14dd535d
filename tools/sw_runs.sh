#!/bin/bash
for r in D fixed R2P; do python bench.py --workload stockwatson --sw-run $r --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$r', d['config']['kernel'], '%.3e'%d['value'], 'tick_us', d.get('tick_us'))
"; done
