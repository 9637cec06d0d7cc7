#!/bin/bash
# rocprofv3 kernel statistics of the bench command, then the plain bench line.  The program follows `--` directly (no env / bash hop).
# usage (GPU box): bash tests/gpu_profile.sh <tag> [bench args, e.g. --config c5]   -> gpurun_out/prof_<tag>*/, gpurun_out/bench_<tag>.log
TAG=${1:-run}
shift
EXTRA="$@"
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --mode both --steps 5 --warmup 2 --train-steps 3 --no-cpu-baseline --no-roofline $EXTRA > $R/gpurun_out/rocprof_$TAG.log 2>&1
echo "rocprof exit $?"
# sampling leg alone: the 3x3x3 k_conv rows of this table are exactly the launches per step that roofline.avg_launch_ms averages
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_sample -- python3 $R/bench.py --mode sample --steps 10 --warmup 3 --no-cpu-baseline --no-roofline $EXTRA > $R/gpurun_out/rocprof_${TAG}_sample.log 2>&1
echo "rocprof (sample) exit $?"
cd $R
timeout -k 10 900 python bench.py --dump-ops gpurun_out/ops_$TAG.txt $EXTRA > gpurun_out/bench_$TAG.log 2>&1
echo "bench exit $?"
tail -c 1500 gpurun_out/bench_$TAG.log
