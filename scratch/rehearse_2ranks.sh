#!/bin/bash
# the two-rank rehearsal of bench.py --tempering on one GPU (gloo), $1 times; stops at the first
# run that does not end by itself (the watchdog then leaves every thread's stack in the .err file)
n=$1; out=gpurun_out/rehearse; mkdir -p $out
for i in $(seq 1 $n); do
  MCMC_AMD_BENCH_WATCHDOG=45 MCMC_AMD_DIST_BACKEND=gloo timeout -k 10 90 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29720 + i)) bench.py --gpus 2 --tempering --chains 512 --steps 200 --warmup 50 > $out/run$i.json 2> $out/run$i.err
  rc=$?
  echo "run $i rc=$rc $(tail -c 120 $out/run$i.json | head -c 60)" | tee -a $out/summary.txt
  if [ $rc -ne 0 ]; then exit 0; fi
done
