#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; R=$PWD
mkdir -p gpurun_out
O=$R/gpurun_out/r06_predict.txt
: > $O
timeout 900 python -m pytest tests/test_gpu_predict.py tests/test_gpu_quat_ops.py tests/test_gpu_parity.py -x -q -k "predict or factor" 2>&1 | tail -5 >> $O
for round in 1 2 3; do
  python tools/microbench.py predict_v1:1000000:0:100 predict_v2:1000000:0:100 predict_v1:100000:0:300 2>&1 | grep -E "launch_us|rror" >> $O
done
cat $O
