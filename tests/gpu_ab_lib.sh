# A/B of two builds of librho_hip.so on one box: the in-tree one, then tools/probe/$1 copied over it (scratch copy on the box only)
mkdir -p gpurun_out
for tag in base alt base2 alt2; do
  if [ "$tag" = "alt" ] || [ "$tag" = "alt2" ]; then cp tools/probe/$1 rho_diffusion_amd/librho_hip.so; else cp gpurun_out/.orig_lib.so rho_diffusion_amd/librho_hip.so 2>/dev/null || cp rho_diffusion_amd/librho_hip.so gpurun_out/.orig_lib.so; fi
  timeout -k 10 300 python bench.py --mode $2 --steps 6 --warmup 2 --train-steps 3 --no-cpu-baseline > gpurun_out/ablib_$tag.log 2>&1
  python - <<PY
import json,re
t=open("gpurun_out/ablib_$tag.log").read()
m=re.search(r'^\{.*\}$', t, re.M)
j=json.loads(m.group(0)); print("$tag", round(j["ms_per_step"],2), j["roofline"]["by_kind_ms"].get("conv3"), (round(j["training"]["ms_per_step"],1) if "training" in j else ""))
PY
done
rm -f gpurun_out/.orig_lib.so
