#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_backward_kernels.py -m gpu -q -p no:cacheprovider -k "wgrad" 2>&1 | tail -3
timeout -k 10 600 python bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench6.log 2>&1; echo "bench exit $?" >> gpurun_out/bench6.log; tail -c 1800 gpurun_out/bench6.log
