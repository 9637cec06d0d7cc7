#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "attention or round3 or backward_fp32 or backward_bf16 or cabi" > gpurun_out/r3b_tests.log 2>&1
echo "exit $?" >> gpurun_out/r3b_tests.log
tail -5 gpurun_out/r3b_tests.log
for c in c2 c1 c5; do
  timeout -k 10 300 python bench.py --config $c --steps 10 --warmup 3 --train-steps 3 --no-cpu-baseline --dump-ops gpurun_out/r3b_ops_$c.txt > gpurun_out/r3b_bench_$c.log 2>&1; echo "bench $c exit $?" >> gpurun_out/r3b_bench_$c.log
  tail -c 300 gpurun_out/r3b_bench_$c.log
done
