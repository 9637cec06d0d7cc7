#!/bin/bash
# same-box A/B of the conv k-split (RHO_CONV_SPLITK=0 disables it) on one bench config: bash tests/gpu_ab_splitk.sh "<bench args>" <tag>
mkdir -p gpurun_out
for tag in on off on2 off2; do
  F=1; if [ "$tag" = "off" ] || [ "$tag" = "off2" ]; then F=0; fi
  RHO_CONV_SPLITK=$F timeout -k 10 400 python bench.py $1 --no-cpu-baseline --no-checkpoint-leg > gpurun_out/absk_$2_$tag.log 2>&1 || { echo "bench failed ($tag)"; tail -5 gpurun_out/absk_$2_$tag.log; exit 1; }
  python - <<PY
import json,re
t=open("gpurun_out/absk_$2_$tag.log").read()
m=re.search(r'^\{.*\}$', t, re.M)
j=json.loads(m.group(0)); tr=j.get("training")
print("$tag splitk=$F", j["metric"], round(j["value"],2), "ms", round(j["ms_per_step"],2), (j.get("roofline") or {}).get("by_kind_ms"),
      "| train", tr and round(tr["value"],1), tr and round(tr["ms_per_step"],1))
PY
done
