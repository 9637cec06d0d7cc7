# same-box A/B of one engine environment switch on the bench, whole per-kind breakdown:
# AB_VAR=<name> AB_ON=<value> AB_OFF=<value> AB_MODE=<train|sample> [AB_ARGS="--config c1"] bash tests/gpu_ab_env2.sh
mkdir -p gpurun_out
VAR=${AB_VAR:-RHO_DW_ARENA}; MODE=${AB_MODE:-train}
for tag in on off on2 off2; do
  F=${AB_ON:-1}; if [ "$tag" = "off" ] || [ "$tag" = "off2" ]; then F=${AB_OFF:-0}; fi
  env $VAR=$F timeout -k 10 400 python bench.py --mode $MODE --steps 6 --warmup 2 --train-steps 6 --no-cpu-baseline --no-checkpoint-leg ${AB_ARGS:-} > gpurun_out/abenv2_$tag.log 2>&1
  python - <<PY
import json,re
t=open("gpurun_out/abenv2_$tag.log").read()
m=re.search(r'^\{.*\}$', t, re.M)
if not m:
    print("$tag $VAR=$F FAILED", t[-800:])
else:
    j=json.loads(m.group(0)); tr=j.get("training")
    if tr:
        bk=tr.get("by_kind_ms", {})
        print("$tag $VAR=$F train", round(tr["ms_per_step"],2), "ms non_mfma", tr["roofline"]["non_mfma_ms"], "bwd", {k: v for k, v in bk.get("bwd", {}).items() if k in ("memset","wgrad","wgrad_finalize","bias_grad","dgrad","add","gn_bwd_apply","attention_bwd")})
    else:
        print("$tag $VAR=$F sample", round(j["ms_per_step"],2), "ms", (j.get("roofline") or {}).get("by_kind_ms"))
PY
done
