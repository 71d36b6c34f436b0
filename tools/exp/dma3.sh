#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
# the measurement switches below exist only in the -DCPI_EXPERIMENTS build (python -m cpi_amd.build --experiments)
export CPI_AMD_LIB=$PWD/cpi_amd/libcpi_amd_exp.so
OUT=gpurun_out/exp_dma3.txt; mkdir -p gpurun_out; : > $OUT
for rep in 1 2; do
for cfg in off 4,2,0 4,2,1; do
  echo "=== rep $rep cfg $cfg" >> $OUT
  CPI_AMD_MEAN_DMA=$cfg timeout 300 python tools/microbench.py v1_mean:1000000:1:30 v1_mean:400000:1:60 v1_mean:200000:1:100 v2_mean:1000000:1:30 2>&1 | grep -v amdgpu.ids >> $OUT
  CPI_MB_SAMPLES=100 CPI_AMD_MEAN_DMA=$cfg timeout 300 python tools/microbench.py v1_mean:1000000:1:15 2>&1 | grep -v amdgpu.ids >> $OUT
done; done
cat $OUT
