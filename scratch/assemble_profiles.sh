#!/bin/bash
# (in the container) turns what scratch/collect_final.sh left under gpurun_out/ into the files
# kept under profiles/
set -e
F=gpurun_out/final_r3
cp $F/bench.json profiles/r3_bench.json
cp $F/bench_steps20_warmup5.json profiles/r3_bench_steps20_warmup5.json
cp $F/bench_tempering.json profiles/r3_bench_tempering.json
cp $F/bench_tempering_thin50.json profiles/r3_bench_tempering_thin50.json
cp $F/bench_tempering_geometric.json profiles/r3_bench_tempering_geometric.json
cp $F/bench_tempering_geometric_thin50.json profiles/r3_bench_tempering_geometric_thin50.json
cp $F/bench_2ranks_gloo.json profiles/r3_bench_2ranks_gloo_one_gpu.json
cp $F/bench_2ranks_gloo_tempering.json profiles/r3_bench_2ranks_gloo_one_gpu_tempering.json
cp gpurun_out/prof_final/bench_under_rocprof.json profiles/r3_bench_under_rocprof.json
{ echo "# bench.py variants, 1000 steps after 200 warm-up, one MI355X (scratch/variants_r3.sh)"; cat $F/variants.txt; } > profiles/r3_variants.txt
{ echo "# trajectories of the chains: bench.py --no-cpu --steps K --warmup W for several K (the draws, and with them the chains' paths, depend on K); MH steps/s"; echo "# long: W = 200"; cat $F/traj_long.txt; python scratch/avg.py $F/traj_long.txt; echo "# short: W = 5"; cat $F/traj_short.txt; python scratch/avg.py $F/traj_short.txt; echo "# the driver's command six times in a row (same chains, same schedule: what is left is the machine)"; cat $F/repeat_steps20.txt; python scratch/avg.py $F/repeat_steps20.txt; } > profiles/r3_trajectories.txt
python tools/rocprof_summary.py gpurun_out/prof_final/kt/bench_results.db profiles/r3_kernel_stats.txt "python3 bench.py --no-cpu (200 warm-up + 2000 timed steps, calls of 500)" >/dev/null
python tools/rocprof_summary.py gpurun_out/prof_final_steps20/kt/bench_results.db profiles/r3_kernel_stats_steps20.txt "python3 bench.py --no-cpu --steps 20 --warmup 5 (the driver's arguments)" >/dev/null
python tools/pmc_summary.py gpurun_out/prof_final profiles/r3_pmc.txt 2200 "" >/dev/null
python tools/pmc_summary.py gpurun_out/prof_final_steps20 profiles/r3_pmc_steps20.txt 25 "--steps 20 --warmup 5" >/dev/null
grep -v amdgpu.ids $F/stage_profile.txt > profiles/r3_stage_profile.txt
grep -v amdgpu.ids $F/stage_profile_steps20.txt > profiles/r3_stage_profile_steps20.txt
{ echo "# scratch/prof_tail.py on the light profiling build (chain start and end times only, full speed): the last 500-step call of bench.py --steps 1000 --warmup 200, then of --steps 2000"; for f in $F/tail_light.txt $F/tail_light_steps2000.txt; do grep -v "amdgpu.ids\|RuntimeWarning\|stddev\|^all:\|^cycles per\|^last chain:\|^  repairs:" $f | sed 's/ repairs\/fwd.*//; s/, repairs per forward.*//'; echo; done; } > profiles/r3_tail.txt
{ echo "# scratch/e2e_rate.py: Inversion.random_walk end to end, 1024 chains, 6 layers, 40 periods, n_thin 10, n_chunk 50, record_diagnostics=step, one MI355X"; grep -v amdgpu.ids $F/e2e.txt; } > profiles/r3_e2e.txt
for f in bench.json bench_steps20_warmup5.json bench_tempering.json bench_tempering_thin50.json bench_tempering_geometric.json bench_tempering_geometric_thin50.json bench_2ranks_gloo.json bench_2ranks_gloo_tempering.json; do python - "$F/$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c=d["config"]; r=d["roofline"]
print(sys.argv[1].split("/")[-1], "%.3f M"%(d["value"]/1e6), "frac %.4f"%r["frac"], "launch_ms %.2f"%r["launch_ms"], c.get("exchanges_accepted_proposed"), d.get("cpu_baseline",{}).get("value"), "exec/lit %.2f"%(r["executed_evals_per_launch"]/r["algorithmic_evals_per_launch"]))
PY
done
