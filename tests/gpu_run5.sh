#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_backward_kernels.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/train.log 2>&1
echo "train exit $?" >> gpurun_out/train.log
tail -40 gpurun_out/train.log
