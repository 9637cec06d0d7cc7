#!/bin/bash
# usage: bash tests/gpu_quick.sh "<pytest -k expr>" <bench mode> <tag>
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "$1" > gpurun_out/tq_$3.log 2>&1; tail -3 gpurun_out/tq_$3.log
timeout -k 10 600 python bench.py --mode $2 --steps 5 --warmup 2 --train-steps 3 --no-cpu-baseline --dump-ops gpurun_out/ops_$3.txt > gpurun_out/bench_$3.log 2>&1; echo "bench exit $?" >> gpurun_out/bench_$3.log
python - <<PY
import json,re
t=open("gpurun_out/bench_$3.log").read()
m=re.search(r'^\{.*\}$', t, re.M)
if m:
    j=json.loads(m.group(0))
    print(j["metric"], round(j["value"],3), "ms", round(j["ms_per_step"],2))
    if "training" in j: print("train", round(j["training"]["value"],2), "samples/s", round(j["training"]["ms_per_step"],1), "ms"); print(j["training"].get("by_kind_ms"))
    if j.get("roofline"): print("conv TF/s", round(j["roofline"]["achieved"],1), j["roofline"].get("by_kind_ms"))
else:
    print(t[-1500:])
PY
