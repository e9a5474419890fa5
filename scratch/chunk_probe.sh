#!/bin/bash
# the length of a call (steps between two returns to the host) against the throughput, over trajectories
for chunk in 250 500 1000 2000 4000; do
  tot=0; n=0
  for k in 2000 2100 2300 2500 3000 4000; do
    v=$(timeout -k 10 300 python bench.py --no-cpu --steps $k --warmup 200 --chunk $chunk 2>/dev/null | python -c "import sys,json; print('%.0f'%json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])") || exit 1
    echo "chunk $chunk steps $k $v"
  done
done
