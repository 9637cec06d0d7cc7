#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_backward_kernels.py -m gpu -q -p no:cacheprovider -x -k "conv" > gpurun_out/t13.log 2>&1; tail -3 gpurun_out/t13.log
timeout -k 10 600 python bench.py --mode both --steps 5 --warmup 2 --train-steps 3 --no-cpu-baseline --dump-ops gpurun_out/ops_f.txt > gpurun_out/bench10.log 2>&1; echo "bench exit $?" >> gpurun_out/bench10.log; tail -c 2400 gpurun_out/bench10.log
