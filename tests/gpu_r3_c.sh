#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "wgrad or backward or bench_geometry or layer_at or round3 or training or updown" > gpurun_out/r3c_tests.log 2>&1
echo "exit $?" >> gpurun_out/r3c_tests.log
tail -5 gpurun_out/r3c_tests.log
for c in c2 c1; do
  timeout -k 10 300 python bench.py --config $c --mode train --steps 5 --warmup 2 --no-cpu-baseline --dump-ops gpurun_out/r3c_ops_$c.txt > gpurun_out/r3c_bench_$c.log 2>&1; echo "bench $c exit $?" >> gpurun_out/r3c_bench_$c.log
  tail -c 400 gpurun_out/r3c_bench_$c.log
done
