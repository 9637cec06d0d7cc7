#!/bin/bash
# PMC passes of the sampling leg (separate --pmc runs: SQ set 1, SQ set 2, FETCH_SIZE, WRITE_SIZE) + a plain kernel trace of the
# same command (per-dispatch durations for the MFMA-utilisation figure).  usage (GPU box): bash tests/gpu_pmc.sh [bench args]
# -> gpurun_out/pmc/{sq1,sq2,fetch,write,trace}, gpurun_out/pmc/meta.json; then (anywhere) python tools/pmc_summary.py <tag>
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
EXTRA="$@"
python3 - <<PY > $R/gpurun_out/pmc/meta.json
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
CMD="python3 $R/bench.py --mode sample --steps 2 --warmup 1 --no-cpu-baseline --no-roofline $EXTRA"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pmc/trace -- $CMD > $R/gpurun_out/pmc/trace.log 2>&1; echo "trace $?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc/sq1 -- $CMD > $R/gpurun_out/pmc/sq1.log 2>&1; echo "sq1 $?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/pmc/sq2 -- $CMD > $R/gpurun_out/pmc/sq2.log 2>&1; echo "sq2 $?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $R/gpurun_out/pmc/grbm -- $CMD > $R/gpurun_out/pmc/grbm.log 2>&1; echo "grbm $?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc/fetch -- $CMD > $R/gpurun_out/pmc/fetch.log 2>&1; echo "fetch $?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc/write -- $CMD > $R/gpurun_out/pmc/write.log 2>&1; echo "write $?"
cd $R; find gpurun_out/pmc -name "*.csv" | head -20; du -sh gpurun_out/pmc
