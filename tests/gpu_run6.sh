#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python bench.py --steps 3 --warmup 1 --train-steps 2 --cpu-steps 1 > gpurun_out/bench3.log 2>&1; echo "bench exit $?" >> gpurun_out/bench3.log; tail -c 3500 gpurun_out/bench3.log
