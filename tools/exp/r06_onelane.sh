#!/bin/bash
# Round 6 (late): work that every lane of a factor repeated, done by the lanes that need it -- the pivot's Newton sequence of the
# square-root-information kernel on lane k, the evaluateError core of cpi_factor_kernel on lane 0 (results through LDS), the Hessian
# sweep's core on lanes 0-4 -- against libcpi_amd_piv16.so (all of it on every lane; otherwise the same tree).  Same instructions per
# wavefront; what changes is how many FP64 lanes switch.  Long runs (300 launches per repetition): the steady state of the clocks.
cd ${GRAFT_REPO_ROOT:-.}; R=$PWD
mkdir -p gpurun_out
O=$R/gpurun_out/r06_onelane.txt
: > $O
timeout 1500 python -m pytest tests/test_gpu_whitening.py tests/test_gpu_packed.py tests/test_gpu_parity.py -x -q 2>&1 | tail -2 >> $O
ROWS="sqrt_info:1000000:0:300 sqrt_info_packed:1000000:0:300 factor_v1_whitened:1000000:0:300 factor_v1_whitened_tri:1000000:0:300 factor_v2_whitened_tri:1000000:0:300 factor_v1:1000000:0:300 factor_v2:1000000:0:300 factor_v1_hessian_tri:1000000:0:300 factor_v1:20000:0:1000 factor_v1_whitened_tri:20000:0:1000"
for round in 1 2 3; do
  for lib in libcpi_amd_piv16.so libcpi_amd.so; do
    CPI_AMD_LIB=$R/cpi_amd/$lib python tools/microbench.py $ROWS 2>&1 | grep launch_us >> $O
  done
done
cat $O
