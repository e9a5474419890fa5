#!/bin/bash
# on the GPU box: the other configurations of the path, 1000 steps each
run() { timeout -k 10 200 python bench.py --no-cpu --steps 1000 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', '|', round(d['value']), 'steps/s frac %.4f' % d['roofline']['frac'])" || exit 1; }
run
run --rotation on
run --itype group
run --wave love
run --layers 12
run --layers 3
run --chains 4096
