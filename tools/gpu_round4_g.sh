#!/bin/bash
# Round-4 session G: assembler with 4 rows per trip (9 wavefronts per CU) vs 8 (4 per CU); dense mean kernel with 3 / 4 knots per
# chunk: time AND FETCH_SIZE; stream / tiled tests on the restored piece assembler.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python -m pytest tests/test_stream.py tests/test_gpu_tiled.py -m gpu -q 2>&1 | tail -30 > gpurun_out/r04_pytest_g.txt; tail -6 gpurun_out/r04_pytest_g.txt
for v in "" _asm4; do CPI_AMD_LIB=$PWD/cpi_amd/libcpi_amd$v.so python tools/microbench.py v1_mean_tiled:1000000:0 v1_mean_tiled:100000:0 2>&1 | tee -a gpurun_out/r04_mb_g.txt; done
for v in "" _c3 _c4; do CPI_AMD_LIB=$PWD/cpi_amd/libcpi_amd$v.so python tools/microbench.py v1_mean:1000000:1 v1_mean:1000000:1 v1_mean_stream:1000000:1 2>&1 | tee -a gpurun_out/r04_mb_g.txt; done
bash tools/exp/pmc_fetch.sh gpurun_out/r04_mean_fetch_ab.txt "cpi_mean_kernel<1" v1_mean:1000000:1:3 cpi_amd/libcpi_amd.so cpi_amd/libcpi_amd_c3.so cpi_amd/libcpi_amd_c4.so
bash tools/exp/pmc_fetch.sh gpurun_out/r04_stream_fetch.txt "cpi_" v1_mean_stream:1000000:0:3 cpi_amd/libcpi_amd.so
bash tools/exp/pmc_fetch.sh gpurun_out/r04_tiled_fetch.txt "cpi_" v1_mean_tiled:1000000:0:3 cpi_amd/libcpi_amd.so
