#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/tests_all.log 2>&1
echo "exit $?" >> gpurun_out/tests_all.log
tail -4 gpurun_out/tests_all.log
