#!/bin/bash
# Round 6, VERDICT r05 item 1: parity of the packed-triangle forms, then their launch times beside the dense rows (two alternating rounds).
cd ${GRAFT_REPO_ROOT:-.}; R=$PWD
mkdir -p gpurun_out
O=$R/gpurun_out/r06_packed.txt
: > $O
timeout 1500 python -m pytest tests/test_gpu_packed.py tests/test_gpu_group.py -x -q -k "not bench" 2>&1 | tail -15 >> $O
for round in 1 2; do
  python tools/microbench.py sqrt_info:1000000:0 sqrt_info_packed:1000000:0 factor_v1_whitened:1000000:0 factor_v1_whitened_tri:1000000:0 \
      factor_v2_whitened:1000000:0 factor_v2_whitened_tri:1000000:0 factor_v1_hessian:1000000:0 factor_v1_hessian_tri:1000000:0 \
      factor_v2_hessian:1000000:0 factor_v2_hessian_tri:1000000:0 factor_v1_packed:1000000:0 factor_v2_packed:1000000:0 predict_v1:1000000:0 predict_v2:1000000:0 2>&1 | grep -E "launch_us|rror" >> $O
done
cat $O
