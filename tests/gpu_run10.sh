#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_backward_kernels.py tests/test_gpu_kernels.py tests/test_gpu_training.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
timeout -k 10 600 python bench.py --mode both --steps 5 --warmup 2 --train-steps 3 --no-cpu-baseline > gpurun_out/bench7.log 2>&1; echo "bench exit $?" >> gpurun_out/bench7.log; tail -c 2300 gpurun_out/bench7.log
