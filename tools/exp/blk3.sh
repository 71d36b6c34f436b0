#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
# the measurement switches below exist only in the -DCPI_EXPERIMENTS build (python -m cpi_amd.build --experiments)
export CPI_AMD_LIB=$PWD/cpi_amd/libcpi_amd_exp.so
mkdir -p gpurun_out
OUT=gpurun_out/exp_blk3.txt
: > $OUT
for L in 8 16; do
  echo "=== check blk $L" >> $OUT
  CPI_AMD_MEAN_BLK=$L timeout 300 python tests/tools/dma_check.py 2>&1 | grep -v amdgpu.ids >> $OUT || echo "CHECK FAILED" >> $OUT
done
for L in 8; do
for WPC in 0 4 6 8; do
for M in 0 1 2; do
  echo "=== blk $L wpc $WPC mode $M" >> $OUT
  CPI_AMD_BLK_WPC=$WPC CPI_AMD_BLK_MODE=$M CPI_AMD_MEAN_BLK=$L timeout 300 python tools/microbench.py v1_mean:1000000:0:20 v1_mean:100000:0:100 v1_mean:10000:0:1000 2>&1 | grep -v amdgpu.ids >> $OUT
done
done
done
echo "=== blk 16 wpc 0 mode 0 (N=50, N=100)" >> $OUT
CPI_AMD_MEAN_BLK=16 timeout 300 python tools/microbench.py v1_mean:1000000:0:20 v1_mean:10000:0:1000 v1_mean:5000:0:1000 2>&1 | grep -v amdgpu.ids >> $OUT
CPI_MB_SAMPLES=100 CPI_AMD_MEAN_BLK=16 timeout 300 python tools/microbench.py v1_mean:1000000:0:10 2>&1 | grep -v amdgpu.ids >> $OUT
CPI_MB_SAMPLES=100 CPI_AMD_MEAN_BLK=8 timeout 300 python tools/microbench.py v1_mean:1000000:0:10 2>&1 | grep -v amdgpu.ids >> $OUT
cat $OUT
