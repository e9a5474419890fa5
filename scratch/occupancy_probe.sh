#!/bin/bash
# what the barrier between a chain's candidates and the speculation cost: the same number of waves with one candidate per chain
run() { timeout -k 10 300 python bench.py --no-cpu --steps 1000 --warmup 200 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%-36s %9.0f steps/s  frac %.4f  exec/lit %.2f  acc %.3f  evals/s %.3e'%('$*', d['value'], r['frac'], r['executed_evals_per_launch']/max(r['algorithmic_evals_per_launch'],1), d['config']['acceptance_rate'], r['executed_evals_per_launch']/r['launch_ms']*1e3))" || exit 1; }
run --chains 1024
run --chains 2048 --candidates 1
run --chains 2048
run --chains 4096 --candidates 1
run --chains 4096
