#!/bin/bash
# Round 6, VERDICT r05 item 3 (first pass, the round-5 kernels as they are): windows of the reference's own lengths (imurate / camrate =
# 10, 20: synthetic_test.launch:27-28) -- launch microseconds by lanes per window, and the copy ceilings of the small sweeps (item 4).
cd ${GRAFT_REPO_ROOT:-.}; R=$PWD
mkdir -p gpurun_out
O=$R/gpurun_out/r06_short_baseline.txt
: > $O
for N in 10 20; do
  echo "== N = $N" >> $O
  for round in 1 2; do
    CPI_MB_SAMPLES=$N python tools/microbench.py v1_mean:1000000:0:40 v1_mean:1000000:1:40 v1_mean:1000000:2:40 \
        v1_mean:100000:0:200 v1_mean:100000:1:200 v1_mean:100000:2:200 v1_mean:100000:3:200 \
        v1_mean:10000:0:1000 v1_mean:10000:1:1000 v1_mean:10000:2:1000 v1_mean:10000:3:1000 v1_mean:10000:4:1000 v1_mean:10000:5:1000 \
        v2_mean:1000000:0:40 v2_mean:10000:0:1000 v2_mean:10000:2:1000 v2_mean:10000:4:1000 \
        v1_mean_tiled:1000000:0:40 v1_mean_tiled:10000:0:1000 v1_mean_stream:1000000:0:40 \
        v1_full:1000000:0:5 v2_full:1000000:0:5 v1_full:10000:0:100 v2_full:10000:0:100 2>&1 | grep -E "launch_us|assembly|Error|error" >> $O
  done
done
echo "== copy ceilings (tools/exp/mix_probe.hip)" >> $O
tools/exp/bin/mix_probe predict64 13824 8192 15625 predict256 55296 32768 3907 \
   packed_v1_21 16296 12096 47620 packed_v2_21 19992 12096 47620 packed_v1_16 12416 9216 62500 packed_v1_32 24832 18432 31250 \
   sqrtinfo_packed4 3840 3840 250000 sqrtinfo_packed16 15360 15360 62500 sqrtinfo_dense4 7200 7200 250000 \
   whitened_tri4 6944 14880 250000 hessian_tri4 6944 15872 250000 whitened_dense4 10304 14880 250000 >> $O 2>&1
cat $O
