#!/bin/bash
# Round 6, VERDICT r05 item 1: parity of the packed-triangle forms, then their launch times beside the dense rows (two alternating rounds).
# usage: tools/exp/r06_packed.sh [tag ...]   (additional cpi_amd/libcpi_amd_<tag>.so variants timed beside the default library)
cd ${GRAFT_REPO_ROOT:-.}; R=$PWD
mkdir -p gpurun_out
O=$R/gpurun_out/r06_packed.txt
: > $O
timeout 1500 python -m pytest tests/test_gpu_packed.py tests/test_gpu_group.py tests/test_gpu_whitening.py -x -q -k "not bench" 2>&1 | tail -15 >> $O
mb() { local lib=cpi_amd/libcpi_amd_$1.so; [ $1 = default ] && lib=cpi_amd/libcpi_amd.so; CPI_AMD_LIB=$R/$lib python tools/microbench.py "${@:2}" 2>&1 | grep -E "launch_us|rror" | sed "s/^/$1 /"; }
for round in 1 2; do for t in default "$@"; do
  mb $t sqrt_info:1000000:0 sqrt_info_packed:1000000:0 factor_v1_whitened:1000000:0 factor_v1_whitened_tri:1000000:0 \
      factor_v2_whitened:1000000:0 factor_v2_whitened_tri:1000000:0 factor_v1_hessian:1000000:0 factor_v1_hessian_tri:1000000:0 \
      factor_v2_hessian:1000000:0 factor_v2_hessian_tri:1000000:0 factor_v1:1000000:0 factor_v1_packed:1000000:0 factor_v2_packed:1000000:0 >> $O
done; done
cat $O
