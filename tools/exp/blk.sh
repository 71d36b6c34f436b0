#!/bin/bash
# GPU experiment: block-resident mean kernel (CPI_AMD_MEAN_BLK=L) -- correctness vs the oracle, then launch times.
cd ${GRAFT_REPO_ROOT:-.}
# the measurement switches below exist only in the -DCPI_EXPERIMENTS build (python -m cpi_amd.build --experiments)
export CPI_AMD_LIB=$PWD/cpi_amd/libcpi_amd_exp.so
mkdir -p gpurun_out
OUT=gpurun_out/exp_blk.txt
: > $OUT
for L in off 4 8 16 32; do
  echo "=== blk $L" >> $OUT
  CPI_AMD_MEAN_BLK=$L timeout 300 python tests/tools/dma_check.py >> $OUT 2>&1 || echo "CHECK FAILED rc=$?" >> $OUT
  CPI_AMD_MEAN_BLK=$L timeout 300 python tools/microbench.py v1_mean:1000000:0:20 v1_mean:100000:0:100 v1_mean:30000:0:300 v1_mean:10000:0:1000 v1_mean:5000:0:1000 v2_mean:1000000:0:20 v2_mean:10000:0:500 2>&1 | grep -v amdgpu.ids >> $OUT
  CPI_MB_SAMPLES=100 CPI_AMD_MEAN_BLK=$L timeout 300 python tools/microbench.py v1_mean:1000000:0:10 2>&1 | grep -v amdgpu.ids >> $OUT
done
cat $OUT
