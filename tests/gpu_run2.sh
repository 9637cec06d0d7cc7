#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_unet.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/unet.log 2>&1
echo "unet exit $?" >> gpurun_out/unet.log
tail -25 gpurun_out/unet.log
timeout -k 10 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log; tail -3 gpurun_out/smoke.log
timeout -k 10 600 python bench.py --steps 3 --warmup 1 --cpu-steps 1 > gpurun_out/bench1.log 2>&1; echo "bench exit $?" >> gpurun_out/bench1.log; tail -5 gpurun_out/bench1.log
