#!/bin/bash
# first-contact GPU script: full diagnostics to gpurun_out/, never stop at the first failure
mkdir -p gpurun_out
rocminfo | grep -m1 gfx > gpurun_out/arch.txt 2>&1
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/kernels.log 2>&1
echo "kernels exit $?" >> gpurun_out/kernels.log
tail -5 gpurun_out/kernels.log
