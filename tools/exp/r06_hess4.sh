#!/bin/bash
# Round 6 (late): the Hessian sweep's table stage compacted to what the record does not hold (9.8 KB with R packed): four wavefronts per SIMD.
# r6a = the tree before the late changes, tri3 = compacted stage but launch bounds of three, default = four.
cd ${GRAFT_REPO_ROOT:-.}; R=$PWD
mkdir -p gpurun_out
O=$R/gpurun_out/r06_hess4.txt
: > $O
timeout 1500 python -m pytest tests/test_gpu_whitening.py tests/test_gpu_packed.py -x -q 2>&1 | tail -3 >> $O
ROWS="factor_v1_hessian:1000000:0 factor_v1_hessian_tri:1000000:0 factor_v2_hessian:1000000:0 factor_v2_hessian_tri:1000000:0 factor_v1_hessian_tri:100000:0:100"
for round in 1 2 3; do
  for lib in libcpi_amd_r6a.so libcpi_amd_e3.so libcpi_amd_f3.so libcpi_amd.so; do
    CPI_AMD_LIB=$R/cpi_amd/$lib python tools/microbench.py $ROWS 2>&1 | grep launch_us >> $O
  done
done
cat $O
