#!/bin/bash
# SQ counter passes of a TRAINING step (two --pmc runs + a plain trace for durations).  usage (GPU box): bash tests/gpu_pmc_train.sh [bench args]
# -> gpurun_out/pmc2/{sq1,sq2,trace}; then: python tools/pmc_summary.py <tag> <steps> pmc2
mkdir -p gpurun_out/pmc2
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
EXTRA="$@"
python3 - <<PY > $R/gpurun_out/pmc2/meta.json
import json, sys
sys.path.insert(0, "$R")
from rho_diffusion_amd import hip
import bench
sys.argv = ["bench.py"] + "$EXTRA".split()
a = bench.parse()
print(json.dumps({"build_id": hip.load().rho_build_info().decode().rsplit("build ", 1)[-1],
                  "workload": dict(dims=a.dims, grid=a.grid, mc=a.mc, batch=a.batch, dtype=a.dtype, labels=a.labels), "extra_args": "$EXTRA"}))
PY
cd /tmp
CMD="python3 $R/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-checkpoint-leg $EXTRA"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pmc2/trace -- $CMD > $R/gpurun_out/pmc2/trace.log 2>&1; echo "trace $?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc2/sq1 -- $CMD > $R/gpurun_out/pmc2/sq1.log 2>&1; echo "sq1 $?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/pmc2/sq2 -- $CMD > $R/gpurun_out/pmc2/sq2.log 2>&1; echo "sq2 $?"
cd $R; du -sh gpurun_out/pmc2
