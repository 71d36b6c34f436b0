#!/bin/bash
# GPU experiment: LDS-DMA mean kernel variants -- correctness vs the oracle, then launch times (run through gpurun).
cd ${GRAFT_REPO_ROOT:-.}
# the measurement switches below exist only in the -DCPI_EXPERIMENTS build (python -m cpi_amd.build --experiments)
export CPI_AMD_LIB=$PWD/cpi_amd/libcpi_amd_exp.so
mkdir -p gpurun_out
OUT=gpurun_out/exp_dma.txt
: > $OUT
for cfg in off 4,2,1 4,3,1 8,1,1 8,2,1 2,2,0 2,3,0 2,4,0 4,2,0 4,3,0; do
  echo "=== cfg $cfg" >> $OUT
  CPI_AMD_MEAN_DMA=$cfg timeout 300 python tests/tools/dma_check.py >> $OUT 2>&1 || echo "CHECK FAILED rc=$?" >> $OUT
  CPI_AMD_MEAN_DMA=$cfg timeout 300 python tools/microbench.py v1_mean:1000000:1:20 v1_mean:100000:1:100 v2_mean:1000000:1:20 >> $OUT 2>&1
  CPI_MB_SAMPLES=100 CPI_AMD_MEAN_DMA=$cfg timeout 300 python tools/microbench.py v1_mean:1000000:1:10 >> $OUT 2>&1
done
cat $OUT
