#!/bin/bash
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
timeout -k 10 600 python bench.py --steps 5 --warmup 2 --dump-ops gpurun_out/ops_r01.txt > gpurun_out/bench2.log 2>&1; echo "bench exit $?" >> gpurun_out/bench2.log; tail -3 gpurun_out/bench2.log
cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; echo "rocprof exit $?"
cd $GRAFT_REPO_ROOT; find gpurun_out/prof -name "*stats*" | head; tail -3 gpurun_out/rocprof.log
