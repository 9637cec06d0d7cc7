#!/bin/bash
# round 3, first GPU call: the whole -m gpu suite, the default bench line, and `bench.py --gpus 2` through the launcher (gloo rehearsal on one GPU)
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r3_tests.log 2>&1
echo "exit $?" >> gpurun_out/r3_tests.log
tail -5 gpurun_out/r3_tests.log
timeout -k 10 400 python bench.py > gpurun_out/r3_bench_c3.log 2>&1; echo "bench exit $?" >> gpurun_out/r3_bench_c3.log
tail -c 600 gpurun_out/r3_bench_c3.log
RHO_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --batch 4 --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > gpurun_out/r3_bench_2rank.log 2> gpurun_out/r3_bench_2rank.err; echo "2rank exit $?" >> gpurun_out/r3_bench_2rank.log
tail -c 900 gpurun_out/r3_bench_2rank.log
