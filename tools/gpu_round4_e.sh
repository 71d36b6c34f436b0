#!/bin/bash
# Round-4 session E: A/B of the mean-kernel staging (lean + 3 waves / lean + 2 waves / round-3 staging), the 16-byte-store assembler,
# cov<2> p rows by symmetry; correctness of the default library on the mean / stream / tiled tests.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python -m pytest tests/test_stream.py tests/test_gpu_tiled.py tests/test_gpu_parity.py tests/test_gpu_group.py::test_bench_gpus_2_end_to_end_rehearsal_on_one_gpu -m gpu -q 2>&1 | tail -60 > gpurun_out/r04_pytest_e.txt; tail -8 gpurun_out/r04_pytest_e.txt
ROWS="v1_mean:10000:0 v1_mean:30000:0 v1_mean:100000:0 v1_mean:1000000:0 v1_mean_stream:1000000:0 v2_mean:10000:0 v2_mean:1000000:0"
python tools/microbench.py $ROWS v1_mean_tiled:1000000:0 v2_full:100000:0 v1_full:100000:0 2>&1 | tee gpurun_out/r04_mb_e.txt
for v in wps2 lean0; do CPI_AMD_LIB=$PWD/cpi_amd/libcpi_amd_$v.so python tools/microbench.py $ROWS 2>&1 | tee -a gpurun_out/r04_mb_e.txt; done
CPI_AMD_LIB=$PWD/cpi_amd/libcpi_amd_psym1.so python tools/microbench.py v2_full:100000:0 v2_full:100000:0 2>&1 | tee -a gpurun_out/r04_mb_e.txt
python tools/microbench.py v2_full:100000:0 v1_mean:10000:0 v1_mean:1000000:0 2>&1 | tee -a gpurun_out/r04_mb_e.txt
