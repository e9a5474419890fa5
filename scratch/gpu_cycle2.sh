#!/bin/bash
# one GPU round: the GPU tests, the stage profile, the bench at the driver's arguments and the default
tag=$1
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/${tag}_t.log 2>&1
rc=$?
tail -3 gpurun_out/${tag}_t.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python scratch/prof_mh.py > gpurun_out/${tag}_prof.log 2>&1 || exit 1
sed -n 2,4p gpurun_out/${tag}_prof.log
timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/${tag}_b20.json 2> gpurun_out/${tag}_b20.err || exit 1
timeout -k 10 200 python bench.py --no-cpu > gpurun_out/${tag}_b2000.json 2> gpurun_out/${tag}_b2000.err || exit 1
python - <<PY
import json
for f in ("gpurun_out/${tag}_b20.json","gpurun_out/${tag}_b2000.json"):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(f, "value %.0f frac %.4f exec/lit %.2f acc %.3f"%(d["value"], r["frac"], r["executed_evals_per_launch"]/r["algorithmic_evals_per_launch"], d["config"]["acceptance_rate"]))
PY
