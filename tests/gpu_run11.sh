#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_backward_kernels.py -m gpu -q -p no:cacheprovider -k "conv" > gpurun_out/t11.log 2>&1; tail -2 gpurun_out/t11.log
timeout -k 10 600 python bench.py --mode both --steps 5 --warmup 2 --train-steps 3 --no-cpu-baseline --dump-ops gpurun_out/ops_d.txt > gpurun_out/bench8.log 2>&1; echo "bench exit $?" >> gpurun_out/bench8.log; tail -c 2300 gpurun_out/bench8.log
