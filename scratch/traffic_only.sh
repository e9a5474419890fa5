#!/bin/bash
# on the GPU box: three bench runs, then the two HBM-traffic PMC passes of one short bench run
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash scratch/bench3.sh || exit 1
rm -rf gpurun_out/tr && mkdir -p gpurun_out/tr
for pass in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 240 rocprofv3 --pmc $pass --output-format csv -d gpurun_out/tr/pmc_$pass -o pmc -- python3 bench.py --no-cpu --steps 500 --warmup 250 > gpurun_out/tr/pmc_$pass.json 2> gpurun_out/tr/pmc_$pass.err || exit 1
done
