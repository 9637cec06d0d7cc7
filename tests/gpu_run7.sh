#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "conv" -p no:cacheprovider 2>&1 | tail -3
timeout -k 10 600 python bench.py --mode sample --steps 5 --warmup 2 --no-cpu-baseline --dump-ops gpurun_out/ops_b.txt > gpurun_out/bench4.log 2>&1; echo "bench exit $?" >> gpurun_out/bench4.log; tail -c 1500 gpurun_out/bench4.log
