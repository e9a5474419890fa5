#!/bin/bash
# everything profiles/ keeps for the round, from the library as built: bench lines, the two-rank
# rehearsal, variants, trajectories (scratch/final_r3.sh), tempering with the geometric ladder,
# kernel trace and counter passes of the default command and of the driver's arguments
# (scratch/profile_r3.sh), stage profile and tail of a call (profiling builds), end-to-end rate
set -o pipefail
out=gpurun_out/final_r3; mkdir -p $out
bash scratch/final_r3.sh 2>&1 | tail -8 || exit 1
timeout -k 10 300 python bench.py --tempering --ladder geometric --no-cpu > $out/bench_tempering_geometric.json 2> $out/geo.err || exit 1
timeout -k 10 300 python bench.py --tempering --ladder geometric --thin 50 --no-cpu > $out/bench_tempering_geometric_thin50.json 2>> $out/geo.err || exit 1
bash scratch/profile_r3.sh prof_final 2>&1 | tail -4 || exit 1
bash scratch/profile_r3.sh prof_final_steps20 --steps 20 --warmup 5 2>&1 | tail -4 || exit 1
MCMC_AMD_LIB=$GRAFT_REPO_ROOT/mcmc_amd/lib/libmcmc_amd_prof.so timeout -k 10 200 python scratch/prof_mh2.py > $out/stage_profile.txt 2>&1 || exit 1
MCMC_AMD_LIB=$GRAFT_REPO_ROOT/mcmc_amd/lib/libmcmc_amd_prof.so timeout -k 10 200 python scratch/prof_short.py > $out/stage_profile_steps20.txt 2>&1 || exit 1
MCMC_AMD_LIB=$GRAFT_REPO_ROOT/mcmc_amd/lib/libmcmc_amd_prof_light.so timeout -k 10 200 python scratch/prof_tail.py > $out/tail_light.txt 2>&1 || exit 1
MCMC_AMD_LIB=$GRAFT_REPO_ROOT/mcmc_amd/lib/libmcmc_amd_prof_light.so timeout -k 10 200 python scratch/prof_tail.py --steps 2000 > $out/tail_light_steps2000.txt 2>&1 || exit 1
timeout -k 10 400 python scratch/e2e_rate.py > $out/e2e.txt 2>&1 || exit 1
echo collected
