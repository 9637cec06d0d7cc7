#!/bin/bash
# usage: bash tests/gpu_quick2.sh "<pytest -k expr>" "<bench args>" <tag>   (tests, then one bench line summarised)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "$1" > gpurun_out/q2_$3_tests.log 2>&1; echo "exit $?" >> gpurun_out/q2_$3_tests.log; tail -4 gpurun_out/q2_$3_tests.log
timeout -k 10 600 python bench.py $2 --no-cpu-baseline --no-checkpoint-leg > gpurun_out/q2_$3_bench.log 2>&1; echo "bench exit $?" >> gpurun_out/q2_$3_bench.log
python - <<PY
import json,re
t=open("gpurun_out/q2_$3_bench.log").read()
m=re.search(r'^\{.*\}$', t, re.M)
if m:
    j=json.loads(m.group(0))
    print(j["metric"], round(j["value"],3), "ms", round(j["ms_per_step"],2))
    tr=j.get("training")
    if tr: print("train", round(tr["value"],2), "samples/s", round(tr["ms_per_step"],1), "ms", (tr.get("by_kind_ms") or {}).get("bwd"))
else:
    print(t[-1500:])
PY
