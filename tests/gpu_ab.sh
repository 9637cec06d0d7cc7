mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py tests/test_gpu_backward_kernels.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -8
for t in 0 1; do
  RHO_CONV_M16=$t timeout -k 10 300 python bench.py --mode both --steps 5 --warmup 2 --train-steps 3 --no-cpu-baseline --dump-ops gpurun_out/ops_m16_$t.txt > gpurun_out/m16_$t.log 2>&1
  python - <<PY
import json,re
t=open("gpurun_out/m16_$t.log").read()
m=re.search(r'^\{.*\}$', t, re.M)
if m:
    j=json.loads(m.group(0)); print("m16 $t", round(j["ms_per_step"],2), j["roofline"]["by_kind_ms"]["conv3"], "train", round(j["training"]["ms_per_step"],1), j["training"]["by_kind_ms"]["fwd"]["conv3"], j["training"]["by_kind_ms"]["bwd"]["dgrad"])
else: print(t[-2000:])
PY
done
