#!/bin/bash
# the measurements kept under profiles/ (r2_*), into gpurun_out/final: bench lines, variants, trajectory averages
set -o pipefail
out=gpurun_out/final; mkdir -p $out
timeout -k 10 400 python bench.py > $out/bench.json 2> $out/bench.err || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $out/bench_steps20_warmup5.json 2>> $out/bench.err || exit 1
timeout -k 10 300 python bench.py --tempering --no-cpu > $out/bench_tempering.json 2>> $out/bench.err || exit 1
bash scratch/variants_r2.sh > $out/variants.txt || exit 1
bash scratch/ab_traj.sh 200 '1500 1700 1900 2000 2100 2300 2500 3000' libmcmc_amd.so > $out/traj_long.txt || exit 1
bash scratch/ab_traj.sh 5 '12 16 18 20 22 24 26 30 36 40' libmcmc_amd.so > $out/traj_short.txt || exit 1
python scratch/avg.py $out/traj_long.txt; python scratch/avg.py $out/traj_short.txt
