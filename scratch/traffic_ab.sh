#!/bin/bash
# FETCH_SIZE + WRITE_SIZE of the MH kernels for library builds: $1 = tag, rest = libraries in
# mcmc_amd/lib.  Each pass is its own run of bench.py --no-cpu --steps 500 --warmup 100.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
for lib in "$@"; do
  out=gpurun_out/$tag/$lib; mkdir -p $out
  for pass in FETCH_SIZE WRITE_SIZE; do
    MCMC_AMD_LIB=$GRAFT_REPO_ROOT/mcmc_amd/lib/$lib timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d $out/pmc_$pass -o pmc -- python3 bench.py --no-cpu --steps 500 --warmup 100 --chunk 250 > $out/pmc_$pass.json 2> $out/pmc_$pass.err || exit 1
  done
  python3 - <<PY
import csv,collections,glob,json
tot=collections.defaultdict(float)
for f in glob.glob("$out/pmc_*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "mh_steps" in r["Kernel_Name"]: tot[r["Counter_Name"]]+=float(r["Counter_Value"])
v=json.loads(open("$out/pmc_FETCH_SIZE.json").read().strip().splitlines()[-1])["value"]
print("$lib", "fetch KB %.4g write KB %.4g -> GB per 250 steps of 1024 chains: %.2f; steps/s under the profiler %.0f"%(tot["FETCH_SIZE"],tot["WRITE_SIZE"],(tot["FETCH_SIZE"]+tot["WRITE_SIZE"])*1024/1e9*250/600, v))
PY
done
