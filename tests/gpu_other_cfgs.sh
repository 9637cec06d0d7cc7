mkdir -p gpurun_out
timeout -k 10 400 python bench.py --mode sample --dims 3 --grid 128 --mc 32 --batch 2 --steps 3 --warmup 1 --no-cpu-baseline --dump-ops gpurun_out/ops_c5.txt > gpurun_out/cfg_c5.log 2>&1; echo "c5 exit $?"
python - <<PY
import json,re
t=open("gpurun_out/cfg_c5.log").read()
m=re.search(r'^\{.*\}$', t, re.M)
j=json.loads(m.group(0)); print(j["value"], j["ms_per_step"], j["roofline"]["by_kind_ms"], j["roofline"]["achieved"])
PY
grep "attention" gpurun_out/ops_c5.txt | head -3
