#!/bin/bash
# Round-4 session B: full GPU suite on the fused-cut library + A/B of the stream entry (fused vs workspace route).
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --maxfail=6 2>&1 | tail -40 > gpurun_out/r04_pytest_b.txt; cat gpurun_out/r04_pytest_b.txt
python tools/microbench.py v1_mean_stream:1000000:0 v1_mean:1000000:0 v1_mean_tiled:1000000:0 v1_mean_stream:100000:0 v1_mean_stream:10000:0 2>&1 | tee gpurun_out/r04_mb_b.txt
CPI_AMD_LIB=$PWD/cpi_amd/libcpi_amd_exp.so CPI_AMD_NO_FUSED_CUT=1 python tools/microbench.py v1_mean_stream:1000000:0 v1_mean_stream:100000:0 v1_mean_stream:10000:0 2>&1 | tee -a gpurun_out/r04_mb_b.txt
CPI_AMD_LIB=$PWD/cpi_amd/libcpi_amd_exp.so python tools/microbench.py v1_mean_stream:1000000:0 2>&1 | tee -a gpurun_out/r04_mb_b.txt
