#!/bin/bash
# the whole -m gpu suite, then the c3 PMC passes on the same (final) binary
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/final_tests.log 2>&1; echo "tests exit $?" | tee -a gpurun_out/final_tests.log
tail -3 gpurun_out/final_tests.log
bash tests/gpu_pmc.sh > gpurun_out/final_pmc.log 2>&1; tail -2 gpurun_out/final_pmc.log
