#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_backward_kernels.py tests/test_gpu_training.py -m gpu -q -p no:cacheprovider -x > gpurun_out/t12.log 2>&1; tail -3 gpurun_out/t12.log
timeout -k 10 600 python bench.py --mode train --steps 5 --warmup 2 --train-steps 3 --no-cpu-baseline --dump-ops gpurun_out/ops_e.txt > gpurun_out/bench9.log 2>&1; echo "bench exit $?" >> gpurun_out/bench9.log; tail -c 1500 gpurun_out/bench9.log
