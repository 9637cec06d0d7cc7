# A/B of two builds of librho_hip.so on one box: the in-tree library vs tools/probe/$1, selected through RHO_HIP_LIB (hip.py) - the
# in-tree file is never touched.  usage (GPU box): bash tests/gpu_ab_lib.sh <probe .so> <bench mode: sample|train|both>
mkdir -p gpurun_out
for tag in base alt base2 alt2; do
  LIB=""
  if [ "$tag" = "alt" ] || [ "$tag" = "alt2" ]; then LIB="$PWD/tools/probe/$1"; fi
  RHO_HIP_LIB=$LIB timeout -k 10 300 python bench.py --mode $2 --steps 6 --warmup 2 --train-steps 3 --no-cpu-baseline > gpurun_out/ablib_$tag.log 2>&1
  python - <<PY
import json,re
t=open("gpurun_out/ablib_$tag.log").read()
m=re.search(r'^\{.*\}$', t, re.M)
j=json.loads(m.group(0)); print("$tag", j["roofline"].get("build_id") if j.get("roofline") else "", round(j["ms_per_step"],2), (j.get("roofline") or {}).get("by_kind_ms", {}).get("conv3"), (round(j["training"]["ms_per_step"],1) if "training" in j else ""))
PY
done
