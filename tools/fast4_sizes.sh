#!/bin/bash
# the headline kernel at several chain counts: tools/fast4_sizes.sh  (steps/s, ms per launch)
for c in 1024 2048 4096; do
  echo "D=100 C=$c"; python3 bench.py --chains $c --no-extras --no-cpu-baseline --steps 3 2>/dev/null | python3 -c 'import sys,json; j=json.loads(sys.stdin.readline()); print(" ", j["value"], j["ms_per_step"])'
done
