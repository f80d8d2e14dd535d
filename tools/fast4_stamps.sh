#!/bin/bash
# stamps of the pair kernel at a chain count: tools/fast4_stamps.sh <out> <chains>
OUT=$1; C=$2
export WN_LIB=stamps
timeout 300 python3 bench.py --no-extras --no-cpu-baseline --steps 1 --warmup 0 --chains $C 2>&1 | grep -E "WN_STAMPS" | sort | uniq -c | sort -rn | head -12 > $OUT
cat $OUT
