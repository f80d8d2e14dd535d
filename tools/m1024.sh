mkdir -p gpurun_out/r04
for d in 100 96 80; do for w in 0 1; do
  if [ $w = 1 ]; then export WN_FAST4_WHOLE=1; else unset WN_FAST4_WHOLE; fi
  echo "D=$d whole=$w"; python bench.py --chains 1024 --dim $d --no-extras --no-cpu-baseline --steps 3 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.readline()); print(j["value"], j["ms_per_step"])'
done; done
unset WN_FAST4_WHOLE
for c in 512 2048; do for w in 0 1; do
  if [ $w = 1 ]; then export WN_FAST4_WHOLE=1; else unset WN_FAST4_WHOLE; fi
  echo "D=100 C=$c whole=$w"; python bench.py --chains $c --no-extras --no-cpu-baseline --steps 3 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.readline()); print(j["value"], j["ms_per_step"])'
done; done
