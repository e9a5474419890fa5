#!/bin/bash
# usage (on the GPU box): bench3.sh <variant.so or ""> : three bench runs, optionally with another build of the library copied over the default one
if [ -n "$1" ]; then cp "$1" mcmc_amd/lib/libmcmc_amd.so || exit 1; fi
for i in 1 2 3; do
  timeout -k 10 120 python bench.py --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['ms_per_step'],4))" || exit 1
done
