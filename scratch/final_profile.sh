#!/bin/bash
# on the GPU box: the bench line, the kernel-trace summary and the PMC passes of the same command
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/v5
timeout -k 10 300 python bench.py > gpurun_out/v5/bench.json 2> gpurun_out/v5/bench.err || exit 1
tail -c 400 gpurun_out/v5/bench.json; echo
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d gpurun_out/v5/kt -o bench -- python3 bench.py --no-cpu > gpurun_out/v5/bench_under_rocprof.json 2> gpurun_out/v5/kt.err || exit 1
find gpurun_out/v5/kt -name "*.db" | head -3
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout -s KILL 240 rocprofv3 --pmc $pass --output-format csv -d gpurun_out/v5/pmc_$name -o pmc -- python3 bench.py --no-cpu --steps 500 --warmup 250 > gpurun_out/v5/pmc_$name.json 2> gpurun_out/v5/pmc_$name.err || exit 1
  find gpurun_out/v5/pmc_$name -name "*counter_collection.csv" | head -2
done
