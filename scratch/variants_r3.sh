#!/bin/bash
# the table of other configurations of the path: bench.py variants, 1000 steps after 200
run() { timeout -k 10 300 python bench.py --no-cpu --steps 1000 --warmup 200 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%-40s %9.0f steps/s  frac %.4f  exec/lit %.2f  acc %.3f'%('$*', d['value'], r['frac'], r['executed_evals_per_launch']/max(r['algorithmic_evals_per_launch'],1), d['config']['acceptance_rate']))" || exit 1; }
run
run --rotation on
run --itype group
run --itype phase+group
run --itype phase+ellipticity
run --itype phase+ellipticity --rotation on
run --wave love
run --layers 12
run --layers 3
run --chains 4096
run --candidates 1
run --search 1 --candidates 1
run --tempering --thin 10
run --tempering --thin 50
