#!/bin/bash
# Round 6 (late): instruction-count cuts in the two sweeps that are bound by it -- quaternion normalisation by a reciprocal square root
# (cpi_math.hpp: CPI_QUAT_RECIP), the packed square-root-information kernel's address selects / running-sum start, the Hessian
# sweep's triangle loads.  A = libcpi_amd_r6a.so (the tree before: build 05c470c1256fcd96), B = the tree.
cd ${GRAFT_REPO_ROOT:-.}; R=$PWD
mkdir -p gpurun_out
O=$R/gpurun_out/r06_recip.txt
: > $O
timeout 1500 python -m pytest tests/test_gpu_whitening.py tests/test_gpu_packed.py tests/test_gpu_parity.py tests/test_gpu_predict.py tests/test_gpu_quat_ops.py -x -q 2>&1 | tail -4 >> $O
ROWS="factor_v1_hessian:1000000:0 factor_v1_hessian_tri:1000000:0 factor_v2_hessian:1000000:0 factor_v2_hessian_tri:1000000:0 sqrt_info:1000000:0 sqrt_info_packed:1000000:0 factor_v1_whitened:1000000:0 factor_v1_whitened_tri:1000000:0 factor_v2_whitened_tri:1000000:0 factor_v1:1000000:0 factor_v1_packed:1000000:0 factor_v2_packed:1000000:0 predict_v1:1000000:0 predict_v2:1000000:0"
for round in 1 2 3; do
  CPI_AMD_LIB=$R/cpi_amd/libcpi_amd_r6a.so python tools/microbench.py $ROWS 2>&1 | grep launch_us >> $O
  python tools/microbench.py $ROWS 2>&1 | grep launch_us >> $O
done
cat $O
