#!/bin/bash
# on the GPU box: tests, three bench runs, the per-phase profile
timeout -k 10 300 python -u -m pytest tests -x -q -m gpu --timeout 120 --timeout-method thread > gpurun_out/tcycle.log 2>&1
rc=$?; echo rc=$rc; tail -1 gpurun_out/tcycle.log
[ $rc -ne 0 ] && exit 1
bash scratch/bench3.sh || exit 1
timeout -k 10 120 python scratch/prof_pair.py 2>&1 | grep "coarse:\|median 3" -A3 | cut -c1-250
