#!/bin/bash
# Round-4 session H: knots per chunk of the one-lane-per-window mean kernel (2 / 3 / 5), register budget (1 or 2 wavefronts per SIMD),
# tile pitch -- same box; HBM traffic (FETCH_SIZE / WRITE_SIZE) of the default and of the best.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
: > gpurun_out/r04_mb_h.txt
for v in "" _c3 _c3w2 _c3p _c3p2 _c5 ""; do CPI_AMD_LIB=$PWD/cpi_amd/libcpi_amd$v.so python tools/microbench.py v1_mean:1000000:1 v1_mean:1000000:1 v1_mean_stream:1000000:1 v1_mean:100000:1 v1_mean:200000:1 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04_mb_h.txt; done
CPI_MB_SAMPLES=100 CPI_AMD_LIB=$PWD/cpi_amd/libcpi_amd.so python tools/microbench.py cfg5_mean:1000000:1 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04_mb_h.txt
CPI_MB_SAMPLES=100 CPI_AMD_LIB=$PWD/cpi_amd/libcpi_amd_c3p.so python tools/microbench.py cfg5_mean:1000000:1 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04_mb_h.txt
bash tools/exp/pmc_fetch.sh gpurun_out/r04_mean_fetch_ab.txt "cpi_mean_kernel<1" v1_mean:1000000:1:3 cpi_amd/libcpi_amd.so cpi_amd/libcpi_amd_c3p.so cpi_amd/libcpi_amd_c5.so
bash tools/exp/pmc_fetch.sh gpurun_out/r04_stream_fetch.txt "cpi_" v1_mean_stream:1000000:0:3 cpi_amd/libcpi_amd.so cpi_amd/libcpi_amd_c3p.so
