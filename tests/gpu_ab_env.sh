# same-box A/B of one engine environment switch on the bench: AB_VAR=<name> AB_ON=<value> AB_OFF=<value> AB_MODE=<train|sample>
# usage (GPU box): AB_VAR=RHO_PHASE_UPSAMPLE_BWD AB_ON=1 AB_OFF=0 AB_MODE=train bash tests/gpu_ab_env.sh
mkdir -p gpurun_out
VAR=${AB_VAR:-RHO_FUSE_GN_BWD}; MODE=${AB_MODE:-train}
for tag in on off on2 off2; do
  F=${AB_ON:-1}; if [ "$tag" = "off" ] || [ "$tag" = "off2" ]; then F=${AB_OFF:-0}; fi
  env $VAR=$F timeout -k 10 300 python bench.py --mode $MODE --steps 4 --warmup 2 --train-steps 4 --no-cpu-baseline > gpurun_out/abenv_$tag.log 2>&1
  python - <<PY
import json,re
t=open("gpurun_out/abenv_$tag.log").read()
m=re.search(r'^\{.*\}$', t, re.M)
j=json.loads(m.group(0)); tr=j.get("training", j)
print("$tag $VAR=$F", round(tr["ms_per_step"],1), {k: v for k, v in (tr.get("by_kind_ms", {}).get("bwd") or (j.get("roofline") or {}).get("by_kind_ms", {})).items() if k in ("dgrad", "wgrad", "conv3", "pool2x", "upsample", "gn_bwd_reduce")})
PY
done
