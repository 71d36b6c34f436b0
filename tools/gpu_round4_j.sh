#!/bin/bash
# Round-4 session J: the line-aligned assembler (carry of the straddling knot) -- every pass geometry, and timing against the
# 448-byte-piece route (CPI_ASM_ALIGNED=0 build) on the same box; read traffic by counters.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python -m pytest tests/test_stream.py tests/test_gpu_tiled.py "tests/test_gpu_parity.py::test_config2_size_launch_geometries_vs_reference_sample" -m gpu -q 2>&1 | tail -40 > gpurun_out/r04_pytest_j.txt; tail -15 gpurun_out/r04_pytest_j.txt
for v in "" _al0 "" _al0; do CPI_AMD_LIB=$PWD/cpi_amd/libcpi_amd$v.so python tools/microbench.py v1_mean_tiled:1000000:0 v1_mean_tiled:100000:0 2>&1 | grep assembly | tee -a gpurun_out/r04_mb_j.txt; done
bash tools/exp/pmc_fetch.sh gpurun_out/r04_asm_fetch.txt "cpi_assemble" v1_mean_tiled:1000000:0:3 cpi_amd/libcpi_amd.so cpi_amd/libcpi_amd_al0.so
