#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / SQ_INSTS_FLAT of the MH kernels for the default bench command
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$1; mkdir -p $out
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_FLAT SQ_INSTS_VALU"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout -s KILL 240 rocprofv3 --pmc $pass --output-format csv -d $out/pmc_$name -o pmc -- python3 bench.py --no-cpu > $out/pmc_$name.json 2> $out/pmc_$name.err || exit 1
done
python - <<PY
import csv,collections,re,glob
tot=collections.defaultdict(float)
for f in glob.glob("$out/pmc_*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "mh_steps" in r["Kernel_Name"]: tot[r["Counter_Name"]]+=float(r["Counter_Value"])
print({k:"%.4g"%v for k,v in tot.items()}, "GB per 250 steps: %.2f"%((tot["FETCH_SIZE"]+tot["WRITE_SIZE"])*1024/1e9*250/2200))
PY
