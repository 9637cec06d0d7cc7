#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -p no:cacheprovider -x -rs > gpurun_out/r3g_tests.log 2>&1
echo "exit $?" >> gpurun_out/r3g_tests.log
tail -8 gpurun_out/r3g_tests.log
python -c "from rho_diffusion_amd import h5io; print('libhdf5 available on the GPU box:', h5io.available())"
