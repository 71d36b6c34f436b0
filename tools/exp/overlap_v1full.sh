#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
# the measurement switches below exist only in the -DCPI_EXPERIMENTS build (python -m cpi_amd.build --experiments)
export CPI_AMD_LIB=$PWD/cpi_amd/libcpi_amd_exp.so
OUT=gpurun_out/exp_overlap_v1full.txt; mkdir -p gpurun_out; : > $OUT
for rep in 1 2; do for NO in 1 0; do
  echo "=== rep $rep CPI_AMD_NO_OVERLAP=$NO" >> $OUT
  CPI_AMD_NO_OVERLAP=$NO timeout 300 python tools/microbench.py v1_full:100000:0:40 v1_full:10000:0:200 v1_full:1000000:0:4 2>&1 | grep -v amdgpu.ids >> $OUT
  CPI_MB_SAMPLES=100 CPI_AMD_NO_OVERLAP=$NO timeout 300 python tools/microbench.py v1_full:1000000:0:3 2>&1 | grep -v amdgpu.ids >> $OUT
done; done
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 >> $OUT
cat $OUT
