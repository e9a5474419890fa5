#!/bin/bash
# A/B of environment settings over several trajectories: $1 = warm-up, $2 = list of step counts
# (quoted), rest = settings "NAME=VALUE" ("-" = none); prints "<setting> <steps> <steps/s>"
w=$1; list=$2; shift; shift
for k in $list; do
  for kv in "$@"; do
    if [ "$kv" = "-" ]; then e=""; else e="$kv"; fi
    env $e timeout -k 10 200 python bench.py --no-cpu --steps $k --warmup $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$kv', $k, '%.0f'%d['value'])" || exit 1
  done
done
