#!/bin/bash
# alternating A/B of library builds on the bench: $1 = bench args (quoted), $2 = repetitions, rest = libraries
args=$1; reps=$2; shift; shift
for rep in $(seq $reps); do
  for lib in "$@"; do
    MCMC_AMD_LIB=$GRAFT_REPO_ROOT/mcmc_amd/lib/$lib timeout -k 10 200 python bench.py --no-cpu $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$args', '%.0f'%d['value'])" || exit 1
  done
done
