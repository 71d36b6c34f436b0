#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
# the measurement switches below exist only in the -DCPI_EXPERIMENTS build (python -m cpi_amd.build --experiments)
export CPI_AMD_LIB=$PWD/cpi_amd/libcpi_amd_exp.so
OUT=gpurun_out/exp_dma2.txt; mkdir -p gpurun_out; : > $OUT
for cfg in 4,2,0 2,3,0 8,1,1; do for M in 0 1 2; do
  echo "=== dma $cfg mode $M" >> $OUT
  CPI_AMD_BLK_MODE=$M CPI_AMD_MEAN_DMA=$cfg timeout 300 python tools/microbench.py v1_mean:1000000:1:20 2>&1 | grep -v amdgpu.ids >> $OUT
done; done
cat $OUT
