#!/bin/bash
mkdir -p gpurun_out/r04
for d in 100 96 80; do for w in 0 1; do
  if [ $w = 1 ]; then RT="--route FAST4_WHOLE"; else RT=""; fi
  echo "D=$d whole=$w"; python bench.py $RT --chains 1024 --dim $d --no-extras --no-cpu-baseline --steps 3 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.readline()); print(j["value"], j["ms_per_step"])'
done; done
for c in 512 2048; do for w in 0 1; do
  if [ $w = 1 ]; then RT="--route FAST4_WHOLE"; else RT=""; fi
  echo "D=100 C=$c whole=$w"; python bench.py $RT --chains $c --no-extras --no-cpu-baseline --steps 3 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.readline()); print(j["value"], j["ms_per_step"])'
done; done
