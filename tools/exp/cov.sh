#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
# the measurement switches below exist only in the -DCPI_EXPERIMENTS build (python -m cpi_amd.build --experiments)
export CPI_AMD_LIB=$PWD/cpi_amd/libcpi_amd_exp.so
mkdir -p gpurun_out
OUT=gpurun_out/exp_cov.txt
: > $OUT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or seeded or nan or edge" 2>&1 | tail -3 >> $OUT
python tools/microbench.py v1_full:100000:0:30 v2_full:100000:0:30 forster_full:100000:0:30 v1_full:10000:0:100 2>&1 | grep -v amdgpu.ids >> $OUT
cat $OUT
