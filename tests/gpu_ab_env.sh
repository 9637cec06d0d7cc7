# same-box A/B of one engine switch (here RHO_FUSE_GN_BWD = $AB_ON vs $AB_OFF) on the training step
mkdir -p gpurun_out
for tag in on off on2 off2; do
  F=${AB_ON:-128}; if [ "$tag" = "off" ] || [ "$tag" = "off2" ]; then F=${AB_OFF:-0}; fi
  RHO_FUSE_GN_BWD=$F timeout -k 10 300 python bench.py --mode train --steps 3 --warmup 1 --train-steps 4 --no-cpu-baseline > gpurun_out/abgnb_$tag.log 2>&1
  python - <<PY
import json,re
t=open("gpurun_out/abgnb_$tag.log").read()
m=re.search(r'^\{.*\}$', t, re.M)
j=json.loads(m.group(0)); tr=j.get("training", j); print("$tag", round(tr["ms_per_step"],1), {k:v for k,v in tr.get("by_kind_ms",{}).get("bwd",{}).items() if k in ("dgrad","gn_bwd_reduce","gn_bwd_finalize","gn_bwd_apply")})
PY
done
