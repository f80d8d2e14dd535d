#!/bin/bash
for r in fixed D R2P; do echo "== $r"; WN_LIB=ostamps python bench.py --workload stockwatson --sw-run $r --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "WN_STAMPS orbit" | tail -2; done
