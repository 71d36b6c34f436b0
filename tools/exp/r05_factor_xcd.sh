#!/bin/bash
# Round 5, VERDICT item 4: XCD-contiguous workgroup -> factor-group mapping of the sweeps (-DCPI_FACTOR_XCD=1) against the default.
# usage (GPU box): tools/exp/r05_factor_xcd.sh <tag> ...   (cpi_amd/libcpi_amd_<tag>.so; "default" = shipped)
cd ${GRAFT_REPO_ROOT:-.}; R=$PWD
mkdir -p gpurun_out
O=$R/gpurun_out/r05_factor_xcd.txt
: > $O
for t in "$@"; do
  [ $t = default ] && continue
  echo "== parity, lib=$t" >> $O
  CPI_AMD_LIB=$R/cpi_amd/libcpi_amd_$t.so timeout 900 python -m pytest tests/test_gpu_whitening.py tests/test_gpu_parity.py -x -q -k "whiten or hessian or factor or sqrt" 2>&1 | tail -2 >> $O
done
mb() { local lib=cpi_amd/libcpi_amd_$1.so; [ $1 = default ] && lib=cpi_amd/libcpi_amd.so; CPI_AMD_LIB=$R/$lib python tools/microbench.py "${@:2}" 2>&1 | grep "launch_us" | sed "s/^/$1 /"; }
for round in 1 2; do for t in "$@"; do
  mb "$t" factor_v1:1000000:0 factor_v2:1000000:0 factor_v1_whitened:1000000:0 factor_v2_whitened:1000000:0 factor_v1_hessian:1000000:0 factor_v2_hessian:1000000:0 sqrt_info:1000000:0 factor_v1_packed:1000000:0 >> $O
done; done
cat $O
