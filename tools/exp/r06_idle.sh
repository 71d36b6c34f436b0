#!/bin/bash
# Round 6 (late): the covariance kernels' idle lanes (1 of 16 in model 1, 5 of 32 in model 2 -- they ran the RK4 recursion as a
# harmless zero column) sit it out (-DCPI_COV_IDLE_OUT=1): fewer FP64 lanes switching in the most FP64-intense kernels of the library.
cd ${GRAFT_REPO_ROOT:-.}; R=$PWD
mkdir -p gpurun_out
O=$R/gpurun_out/r06_idle.txt
: > $O
CPI_AMD_LIB=$R/cpi_amd/libcpi_amd_idle1.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_packed.py tests/test_gpu_short_windows.py -x -q 2>&1 | tail -2 >> $O
ROWS="v2_full:100000:0:100 v1_full:100000:0:200 v2_full_sym:100000:0:100"
for round in 1 2 3; do
  for lib in libcpi_amd.so libcpi_amd_idle1.so; do
    CPI_AMD_LIB=$R/cpi_amd/$lib python tools/microbench.py $ROWS 2>&1 | grep launch_us >> $O
  done
done
cat $O
