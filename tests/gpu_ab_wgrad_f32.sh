#!/bin/bash
# timing probes of the exact-f32 weight gradient on c2's training step: the shipped library vs builds that skip the reduction
# (NOMMA) or the tile staging (NODMA).  Wrong results by construction: the training loss only has to stay finite.
mkdir -p gpurun_out
for v in tree NOMMA NODMA; do
  if [ $v = tree ]; then unset RHO_HIP_LIB; else export RHO_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab_libs/libwgrad_$v.so; fi
  timeout -k 10 300 python bench.py --config c2 --mode train --steps 4 --warmup 2 --no-cpu-baseline --no-checkpoint-leg > gpurun_out/abw_$v.log 2>&1; echo "$v exit $?"
  python - <<PY
import json,re
t=open("gpurun_out/abw_$v.log").read()
m=re.search(r'^\{.*\}$', t, re.M)
if m:
    j=json.loads(m.group(0)); b=j["training"]["by_kind_ms"]["bwd"]
    print("$v", "step", round(j["ms_per_step"],1), "ms; wgrad", b.get("wgrad"))
else: print(t[-600:])
PY
done
