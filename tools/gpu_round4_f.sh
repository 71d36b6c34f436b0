#!/bin/bash
# Round-4 session F: linear-span assembler (correctness on every pass geometry + timing), cov<2> p rows by symmetry (parity on the
# variant library, counter table before / after).
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python -m pytest tests/test_stream.py tests/test_gpu_tiled.py "tests/test_gpu_parity.py::test_config2_size_launch_geometries_vs_reference_sample" tests/test_gpu_group.py::test_bench_gpus_2_end_to_end_rehearsal_on_one_gpu -m gpu -q 2>&1 | tail -60 > gpurun_out/r04_pytest_f.txt; tail -12 gpurun_out/r04_pytest_f.txt
python tools/microbench.py v1_mean_tiled:1000000:0 v2_mean_tiled:1000000:0 v1_mean_tiled:10000:0 v1_mean_stream:1000000:0 v1_mean:10000:0 v1_mean:1000000:0 2>&1 | tee gpurun_out/r04_mb_f.txt
CPI_AMD_LIB=$PWD/cpi_amd/libcpi_amd_psym1.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_forster.py -m gpu -q 2>&1 | tail -5 | tee gpurun_out/r04_pytest_psym1.txt
bash tools/exp/pmc_ab.sh gpurun_out/r04_cov2_pmc_ab.txt "cpi_cov_kernel<2" v2_full:100000:0:3 cpi_amd/libcpi_amd.so cpi_amd/libcpi_amd_psym1.so | tail -80
