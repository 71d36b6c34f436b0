#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
# the measurement switches below exist only in the -DCPI_EXPERIMENTS build (python -m cpi_amd.build --experiments)
export CPI_AMD_LIB=$PWD/cpi_amd/libcpi_amd_exp.so
OUT=gpurun_out/exp_dma4.txt; mkdir -p gpurun_out; : > $OUT
for cfg in off 4,1,0 4,1,1 6,1,0 6,2,0 4,2,0; do
  echo "=== cfg $cfg" >> $OUT
  CPI_AMD_MEAN_DMA=$cfg timeout 300 python tests/tools/dma_check.py 2>&1 | grep -v amdgpu.ids >> $OUT || echo "CHECK FAILED" >> $OUT
  for M in 0 1 2; do
  CPI_AMD_BLK_MODE=$M CPI_AMD_MEAN_DMA=$cfg timeout 300 python tools/microbench.py v1_mean:1000000:1:30 v1_mean:200000:1:100 2>&1 | grep -v amdgpu.ids | sed "s/^default/mode$M/" >> $OUT
  done
done
cat $OUT
