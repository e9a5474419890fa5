#!/bin/bash
# A/B of library builds over several trajectories (the draws depend on the total number of steps):
# $1 = warm-up, $2 = list of step counts (quoted), rest = libraries
w=$1; list=$2; shift; shift
for k in $list; do
  for lib in "$@"; do
    MCMC_AMD_LIB=$GRAFT_REPO_ROOT/mcmc_amd/lib/$lib timeout -k 10 200 python bench.py --no-cpu --steps $k --warmup $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', $k, '%.0f'%d['value'])" || exit 1
  done
done
