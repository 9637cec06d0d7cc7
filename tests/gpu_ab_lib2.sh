#!/bin/bash
# A/B of the in-tree library against tools/ab_libs/$1 on one box: bash tests/gpu_ab_lib2.sh <probe .so> "<bench args>" "<pytest -k expr for the probe>"
mkdir -p gpurun_out
LIB=$GRAFT_REPO_ROOT/tools/ab_libs/$1
for round in 1 2; do
  for v in tree probe; do
    if [ $v = tree ]; then unset RHO_HIP_LIB; else export RHO_HIP_LIB=$LIB; fi
    timeout -k 10 300 python bench.py $2 --no-cpu-baseline --no-checkpoint-leg > gpurun_out/ab2_${v}_$round.log 2>&1
    python - <<PY
import json,re
t=open("gpurun_out/ab2_${v}_$round.log").read()
m=re.search(r'^\{.*\}$', t, re.M)
if m:
    j=json.loads(m.group(0)); r=j.get("roofline") or {}
    print("$v $round", j["metric"], round(j["value"],3), "ms", round(j["ms_per_step"],2), r.get("by_kind_ms"))
else: print("$v", t[-600:])
PY
  done
done
if [ -n "$3" ]; then
  RHO_HIP_LIB=$LIB timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "$3" > gpurun_out/ab2_tests.log 2>&1; echo "probe tests exit $?"; tail -3 gpurun_out/ab2_tests.log
fi
