#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ddp.py -m gpu -q -p no:cacheprovider --tb=short > gpurun_out/ddp.log 2>&1; tail -15 gpurun_out/ddp.log
# rehearsal of the driver's multi-GPU launch line with 2 ranks sharing the one GPU (gloo), small shape
RHO_DIST_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --grid 16 --batch 4 --no-cpu-baseline > gpurun_out/bench_2rank.log 2>&1; echo "2rank exit $?" >> gpurun_out/bench_2rank.log; tail -c 1500 gpurun_out/bench_2rank.log
