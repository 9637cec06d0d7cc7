#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "attention" > gpurun_out/r3e_tests.log 2>&1
echo "exit $?" >> gpurun_out/r3e_tests.log
tail -4 gpurun_out/r3e_tests.log
timeout -k 10 300 python bench.py --config c2 --mode train --steps 5 --warmup 2 --no-cpu-baseline --no-checkpoint-leg > gpurun_out/r3e_bench_c2.log 2>&1; echo "bench c2 exit $?" >> gpurun_out/r3e_bench_c2.log
tail -c 600 gpurun_out/r3e_bench_c2.log
AB_EXTRA=tools/ab_libs/libconv_gb10.so timeout -k 10 300 python tools/ab_conv.py 32 > gpurun_out/r3e_ab_plain.log 2>&1; echo "ab exit $?" >> gpurun_out/r3e_ab_plain.log
AB_FEATS=sr AB_EXTRA=tools/ab_libs/libconv_gb10.so timeout -k 10 300 python tools/ab_conv.py 32 > gpurun_out/r3e_ab_sr.log 2>&1; echo "ab exit $?" >> gpurun_out/r3e_ab_sr.log
cat gpurun_out/r3e_ab_plain.log gpurun_out/r3e_ab_sr.log
