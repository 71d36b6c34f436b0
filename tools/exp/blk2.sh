#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
# the measurement switches below exist only in the -DCPI_EXPERIMENTS build (python -m cpi_amd.build --experiments)
export CPI_AMD_LIB=$PWD/cpi_amd/libcpi_amd_exp.so
mkdir -p gpurun_out
OUT=gpurun_out/exp_blk2.txt
: > $OUT
for L in 8 16; do
for M in 0 1 2 4; do
  echo "=== blk $L mode $M" >> $OUT
  CPI_AMD_BLK_MODE=$M CPI_AMD_MEAN_BLK=$L timeout 300 python tools/microbench.py v1_mean:1000000:0:20 v1_mean:100000:0:100 v1_mean:10000:0:1000 2>&1 | grep -v amdgpu.ids >> $OUT
done
done
cat $OUT
