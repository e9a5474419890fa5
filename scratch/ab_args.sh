#!/bin/bash
# A/B of library builds for one set of bench arguments over several trajectories: $1 = quoted extra
# bench arguments, $2 = warm-up, $3 = list of step counts (quoted), rest = libraries
extra=$1; w=$2; list=$3; shift; shift; shift
for k in $list; do
  for lib in "$@"; do
    MCMC_AMD_LIB=$GRAFT_REPO_ROOT/mcmc_amd/lib/$lib timeout -k 10 200 python bench.py --no-cpu --steps $k --warmup $w $extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', $k, '%.0f'%d['value'])" || exit 1
  done
done
