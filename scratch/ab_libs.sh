#!/bin/bash
# A/B of library builds on the bench: $1 = bench args (quoted), rest = library files under mcmc_amd/lib
args=$1; shift
for lib in "$@"; do
  for rep in 1 2; do
    MCMC_AMD_LIB=$GRAFT_REPO_ROOT/mcmc_amd/lib/$lib timeout -k 10 200 python bench.py --no-cpu $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$args', '%.0f'%d['value'])" || exit 1
  done
done
