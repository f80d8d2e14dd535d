#!/bin/bash
for v in nost nost1; do WN_LIB=$v python bench.py --workload stockwatson --sw-run fixed --steps 2 --warmup 1 --iters 4 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', d['config']['kernel'], '%.3e'%d['value'], 'tick_us', d.get('tick_us'))
"; done
