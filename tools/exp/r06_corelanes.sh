#!/bin/bash
# Round 6 (late): the Hessian sweep's column-independent core on the five lanes of a factor that use its results (default) against
# all sixteen (-DCPI_HESS_CORE_LANES=16): the same instructions per wavefront, fewer FP64 lanes switching -- does the power controller
# let the sweep run faster in the steady state?  Long runs (300 launches per repetition), alternating.
cd ${GRAFT_REPO_ROOT:-.}; R=$PWD
mkdir -p gpurun_out
O=$R/gpurun_out/r06_corelanes.txt
: > $O
timeout 900 python -m pytest tests/test_gpu_whitening.py tests/test_gpu_packed.py -x -q 2>&1 | tail -2 >> $O
ROWS="factor_v1_hessian_tri:1000000:0:300 factor_v2_hessian_tri:1000000:0:300 factor_v1_hessian:1000000:0:300"
for round in 1 2 3; do
  for lib in libcpi_amd_core16.so libcpi_amd.so; do
    CPI_AMD_LIB=$R/cpi_amd/$lib python tools/microbench.py $ROWS 2>&1 | grep launch_us >> $O
  done
done
cat $O
