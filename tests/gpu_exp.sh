#!/bin/bash
mkdir -p gpurun_out
RHO_EXP_NOPRE=1 timeout -k 10 300 python bench.py --mode sample --steps 3 --warmup 1 --no-cpu-baseline --dump-ops gpurun_out/ops_nopre.txt > gpurun_out/exp_nopre.log 2>&1; grep -o '"by_kind_ms": {[^}]*}' gpurun_out/exp_nopre.log | tail -1
RHO_EXP_NOSILU=1 timeout -k 10 300 python bench.py --mode sample --steps 3 --warmup 1 --no-cpu-baseline --dump-ops gpurun_out/ops_nosilu.txt > gpurun_out/exp_nosilu.log 2>&1; grep -o '"by_kind_ms": {[^}]*}' gpurun_out/exp_nosilu.log | tail -1
