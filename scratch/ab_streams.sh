#!/bin/bash
# on the GPU box: for each named build (scratch/libs/<name>.so) two runs of the bench's stream of
# draws and the per-call times of the 1200-step stream (the one with the outlier chain)
cp mcmc_amd/lib/libmcmc_amd.so /tmp/keep.so
for v in "$@"; do
  cp scratch/libs/$v.so mcmc_amd/lib/libmcmc_amd.so || exit 1
  for i in 1 2; do
    timeout -k 10 120 python bench.py --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v bench', round(d['value']), round(d['ms_per_step'],4))" || exit 1
  done
  timeout -k 10 200 python scratch/per_call.py 1200 2>&1 | grep "^steps" | cut -c1-45 | tr '\n' ' ' || exit 1
  echo
done
cp /tmp/keep.so mcmc_amd/lib/libmcmc_amd.so
