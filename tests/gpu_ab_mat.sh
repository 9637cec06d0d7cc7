mkdir -p gpurun_out
for i in 1 2; do
for v in 100000 512 256; do
RHO_MATERIALIZE_MIN_COUT=$v timeout -k 10 200 python bench.py --mode sample --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_mat_$v.log 2>&1
python - <<PY
import json,re
t=open("gpurun_out/ab_mat_$v.log").read()
m=re.search(r'^\{.*\}$', t, re.M)
j=json.loads(m.group(0)); print("min_cout=$v", round(j["ms_per_step"],2), j["roofline"]["by_kind_ms"])
PY
done
done
