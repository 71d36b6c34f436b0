#!/bin/bash
# Round 6, VERDICT r05 item 4: the packed evaluateError sweep with its result overlaying the dead part of the record (LDS per factor 1 504 -> 928 B):
# parity, then lanes per factor (experiments build: CPI_AMD_PACKED_LPF).
cd ${GRAFT_REPO_ROOT:-.}; R=$PWD
mkdir -p gpurun_out
O=$R/gpurun_out/r06_packed_sweep.txt
: > $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_packed.py tests/test_gpu_whitening.py tests/test_gpu_quat_ops.py -x -q -k "packed or factor or hessian or whiten" 2>&1 | tail -4 >> $O
for round in 1 2; do
  python tools/microbench.py factor_v1_packed:1000000:0 factor_v2_packed:1000000:0 factor_v1_packed:100000:0:200 2>&1 | grep -E "launch_us|rror" >> $O
  for lpf in 2 3 4 6; do
    CPI_AMD_LIB=$R/cpi_amd/libcpi_amd_exp.so CPI_AMD_PACKED_LPF=$lpf python tools/microbench.py factor_v1_packed:1000000:0 factor_v2_packed:1000000:0 factor_v1_packed:100000:0:200 factor_v1_packed:20000:0:500 2>&1 | grep -E "launch_us|rror" | sed "s/^/lpf=$lpf /" >> $O
  done
done
cat $O
