#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_backward_kernels.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/bwd.log 2>&1
echo "bwd exit $?" >> gpurun_out/bwd.log
grep -E "passed|failed|FAILED|Error|error" gpurun_out/bwd.log | head -60
