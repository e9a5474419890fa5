#!/bin/bash
# A/B of library builds for several numbers of chains: $1 = warm-up, $2 = steps, $3 = list of chain counts (quoted), rest = libraries
w=$1; k=$2; list=$3; shift; shift; shift
for c in $list; do
  for lib in "$@"; do
    MCMC_AMD_LIB=$GRAFT_REPO_ROOT/mcmc_amd/lib/$lib timeout -k 10 200 python bench.py --no-cpu --steps $k --warmup $w --chains $c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', $c, '%.0f'%d['value'], 'exec/lit %.2f'%(d['roofline']['executed_evals_per_launch']/d['roofline']['algorithmic_evals_per_launch']))" || exit 1
  done
done
