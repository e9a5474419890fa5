#!/bin/bash
# on the GPU box: alternate bench runs of several builds of the library (scratch/libs/<name>.so), 3 rounds
for i in 1 2 3; do
  for v in "$@"; do
    cp scratch/libs/$v.so mcmc_amd/lib/libmcmc_amd.so || exit 1
    timeout -k 10 120 python bench.py --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', round(d['value']), round(d['ms_per_step'],4), 'exec/lit %.3f'%(r['executed_evals_per_launch']/r['algorithmic_evals_per_launch']))" || exit 1
  done
done
