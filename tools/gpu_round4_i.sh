#!/bin/bash
# Round-4 session I: where the three-knots-per-chunk (BIG) instantiation starts to pay (models 1 and 2), then the GPU suite on the
# default library.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
: > gpurun_out/r04_mb_i.txt
for v in _bigoff _bigall _bigoff _bigall; do CPI_AMD_LIB=$PWD/cpi_amd/libcpi_amd$v.so python tools/microbench.py v1_mean:300000:1 v1_mean:400000:1 v1_mean:500000:1 v1_mean:700000:1 v1_mean:1000000:1 v2_mean:500000:1 v2_mean:1000000:1 v1_mean_stream:1000000:1 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04_mb_i.txt; done
python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee gpurun_out/r04_pytest_i.txt
