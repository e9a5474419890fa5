#!/bin/bash
# the measurements kept under profiles/ (r3_*), into gpurun_out/final_r3: bench lines, the
# two-rank rehearsal (both ranks on the one GPU, gloo standing in for RCCL), variants, trajectory
# averages, repeats of the driver's command
set -o pipefail
out=gpurun_out/final_r3; mkdir -p $out
timeout -k 10 400 python bench.py > $out/bench.json 2> $out/bench.err || exit 1
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_steps20_warmup5.json 2>> $out/bench.err || exit 1
timeout -k 10 300 python bench.py --tempering --no-cpu > $out/bench_tempering.json 2>> $out/bench.err || exit 1
timeout -k 10 300 python bench.py --tempering --thin 50 --no-cpu > $out/bench_tempering_thin50.json 2>> $out/bench.err || exit 1
MCMC_AMD_BENCH_WATCHDOG=60 MCMC_AMD_DIST_BACKEND=gloo timeout -k 10 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 2 --steps 20 --warmup 5 > $out/bench_2ranks_gloo.json 2> $out/bench_2ranks.err || exit 1
MCMC_AMD_BENCH_WATCHDOG=60 MCMC_AMD_DIST_BACKEND=gloo timeout -k 10 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29712 bench.py --gpus 2 --tempering --chains 512 --steps 200 --warmup 50 > $out/bench_2ranks_gloo_tempering.json 2>> $out/bench_2ranks.err || exit 1
echo variants; bash scratch/variants_r3.sh | tee $out/variants.txt || exit 1
bash scratch/ab_traj.sh 200 '1500 1700 1900 2000 2100 2300 2500 3000' libmcmc_amd.so | tee $out/traj_long.txt || exit 1
bash scratch/ab_traj.sh 5 '12 16 18 20 22 24 26 30 36 40' libmcmc_amd.so | tee $out/traj_short.txt || exit 1
bash scratch/ab_traj.sh 5 '20 20 20 20 20 20' libmcmc_amd.so | tee $out/repeat_steps20.txt || exit 1
python scratch/avg.py $out/traj_long.txt; python scratch/avg.py $out/traj_short.txt; python scratch/avg.py $out/repeat_steps20.txt
