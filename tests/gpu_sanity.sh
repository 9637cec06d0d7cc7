#!/bin/bash
# smoke(), the bench under the driver's N = 1 command, `bench.py --gpus 2` through the launcher (gloo rehearsal on one GPU) and the RCCL code path with one rank
mkdir -p gpurun_out
timeout -k 10 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/sanity_smoke.log 2>&1; echo "smoke exit $?" | tee -a gpurun_out/sanity_smoke.log
tail -2 gpurun_out/sanity_smoke.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/sanity_bench_driver_cmd.log 2>&1; echo "bench exit $?" | tee -a gpurun_out/sanity_bench_driver_cmd.log
tail -c 300 gpurun_out/sanity_bench_driver_cmd.log
RHO_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --batch 4 --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > gpurun_out/sanity_bench_2rank.log 2> gpurun_out/sanity_bench_2rank.err; echo "2rank exit $?" | tee -a gpurun_out/sanity_bench_2rank.log
RHO_BENCH_FORCE_DIST=1 timeout -k 10 400 python bench.py --gpus 1 --batch 4 --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > gpurun_out/sanity_bench_rccl1.log 2>&1; echo "rccl world-1 exit $?" | tee -a gpurun_out/sanity_bench_rccl1.log
