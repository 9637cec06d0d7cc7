#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_backward_kernels.py -m gpu -q -p no:cacheprovider -x -k "attention or attn" > gpurun_out/t14.log 2>&1; tail -3 gpurun_out/t14.log
timeout -k 10 600 python bench.py --mode train --steps 5 --warmup 2 --train-steps 3 --no-cpu-baseline --dump-ops gpurun_out/ops_g.txt > gpurun_out/bench11.log 2>&1; echo "bench exit $?" >> gpurun_out/bench11.log; grep -o '"training": {.*"bwd_TFLOPs": {[^}]*}' gpurun_out/bench11.log
