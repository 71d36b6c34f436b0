#!/bin/bash
# Round 6 (late): square-root information with the two trailing updates of a step under complementary lane masks (-DCPI_SQRT_SPLIT_MASK=1:
# lanes j >= k update their running sums, lanes j <= k their column; lane k is the DPP source of both) against the shipped kernel.
cd ${GRAFT_REPO_ROOT:-.}; R=$PWD
mkdir -p gpurun_out; O=$R/gpurun_out/r06_sqrt_split.txt; : > $O
CPI_AMD_LIB=$R/cpi_amd/libcpi_amd_split1.so timeout 900 python -m pytest tests/test_gpu_whitening.py tests/test_gpu_packed.py -x -q 2>&1 | tail -2 >> $O
ROWS="sqrt_info_packed:1000000:0:300 sqrt_info:1000000:0:300 sqrt_info_packed:100000:0:1000"
for round in 1 2 3; do
  for lib in libcpi_amd.so libcpi_amd_split1.so; do
    CPI_AMD_LIB=$R/cpi_amd/$lib python tools/microbench.py $ROWS 2>&1 | grep launch_us >> $O
  done
done
cat $O
