#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "unet_v1 or legacy or groupnorm_any or act_add or round3 or layer_at" > gpurun_out/r3d_tests.log 2>&1
echo "exit $?" >> gpurun_out/r3d_tests.log
tail -5 gpurun_out/r3d_tests.log
grep "split-rounding" gpurun_out/r3d_tests.log | head
