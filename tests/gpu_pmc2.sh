#!/bin/bash
# SQ counters for the training step (wgrad / dgrad / attention backward)
mkdir -p gpurun_out/pmc2
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
CMD="python3 $R/bench.py --mode train --steps 2 --warmup 1 --train-steps 1 --no-cpu-baseline --no-roofline"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc2/sq1 -- $CMD > $R/gpurun_out/pmc2/sq1.log 2>&1; echo "sq1 $?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/pmc2/sq2 -- $CMD > $R/gpurun_out/pmc2/sq2.log 2>&1; echo "sq2 $?"
cd $R; find gpurun_out/pmc2 -name "*.csv" | head -20; du -sh gpurun_out/pmc2
