#!/bin/bash
# on the GPU box: kernel-trace summary and PMC passes of the bench command; $1 = tag, $2.. = bench args
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $out/kt -o bench -- python3 bench.py --no-cpu "$@" > $out/bench_under_rocprof.json 2> $out/kt.err || exit 1
find $out/kt -name "*stats*" | head
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout -s KILL 240 rocprofv3 --pmc $pass --output-format csv -d $out/pmc_$name -o pmc -- python3 bench.py --no-cpu "$@" > $out/pmc_$name.json 2> $out/pmc_$name.err || exit 1
  find $out/pmc_$name -name "*counter_collection.csv" | head -2
done
