#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "attention or wgrad or backward_kernels or layer_at" > gpurun_out/r3f_tests.log 2>&1
echo "exit $?" >> gpurun_out/r3f_tests.log
tail -4 gpurun_out/r3f_tests.log
timeout -k 10 300 python bench.py --config c2 --mode train --steps 5 --warmup 2 --no-cpu-baseline --no-checkpoint-leg > gpurun_out/r3f_bench_c2.log 2>&1; echo "bench c2 exit $?" >> gpurun_out/r3f_bench_c2.log
timeout -k 10 300 python bench.py --config c5 --mode train --steps 5 --warmup 2 --no-cpu-baseline --no-checkpoint-leg > gpurun_out/r3f_bench_c5_fused.log 2>&1; echo "exit $?" >> gpurun_out/r3f_bench_c5_fused.log
RHO_ATTN_DKV_SPLIT=1 timeout -k 10 300 python bench.py --config c5 --mode train --steps 5 --warmup 2 --no-cpu-baseline --no-checkpoint-leg > gpurun_out/r3f_bench_c5_split.log 2>&1; echo "exit $?" >> gpurun_out/r3f_bench_c5_split.log
python - <<'PY'
import json,re
for f in ("r3f_bench_c2","r3f_bench_c5_fused","r3f_bench_c5_split"):
    t=open(f"gpurun_out/{f}.log").read()
    m=re.search(r'^\{.*\}$', t, re.M)
    if not m: print(f, t[-500:]); continue
    j=json.loads(m.group(0))
    b=j["training"]["by_kind_ms"]["bwd"]
    print(f, round(j["value"],2), "samples/s", round(j["ms_per_step"],1), "ms; wgrad", b.get("wgrad"), "attention_bwd", b.get("attention_bwd"))
PY
