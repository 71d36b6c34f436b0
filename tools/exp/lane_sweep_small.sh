#!/bin/bash
# Lanes per window around the headline size: microseconds per launch for L = 4, 5, 6, 8 and the automatic choice (L = 0).
cd ${GRAFT_REPO_ROOT:-.}
for round in 1 2 3; do
  python tools/microbench.py v1_mean:10000:0:600 v1_mean:10000:5:600 v1_mean:10000:6:600 v1_mean:10000:4:600 v1_mean:8000:0:600 v1_mean:8000:5:600 v1_mean:8000:6:600 v1_mean:8000:8:600 \
     v1_mean:12000:0:500 v1_mean:12000:4:500 v1_mean:12000:5:500 v1_mean:12000:6:500 v1_mean:14000:0:500 v1_mean:14000:4:500 v1_mean:14000:5:500 v1_mean:16000:0:500 v1_mean:16000:3:500 v1_mean:16000:4:500 v1_mean:16000:5:500 \
     v2_mean:10000:0:500 v2_mean:10000:5:500 v2_mean:10000:6:500 v2_mean:10000:8:500 2>&1 | grep launch_us
done | tee gpurun_out/r04_lane_sweep_small.txt
