#!/bin/bash
# BIG (three knots per chunk, 32-bit staging offsets) against the two-knot kernel at small one-lane batches: where should it start?
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r04_big2_thresh.txt
: > $O
mb() { local lib=cpi_amd/libcpi_amd${1:+_$1}.so; CPI_AMD_LIB=$PWD/$lib python tools/microbench.py "${@:2}" 2>&1 | grep "launch_us"; }
for round in 1 2; do
  for t in r4base bigall; do
    mb "$t" v1_mean:40000:1:50 v1_mean:65000:1:50 v1_mean:100000:1:30 v1_mean:300000:1 v2_mean:40000:1:50 v2_mean:65000:1:50 v2_mean:100000:1:30 v2_mean:300000:1 v2_mean:500000:1 \
            v1_mean_stream:40000:1:50 v1_mean_stream:65000:1:50 v1_mean_stream:100000:1:30 v1_mean_stream:300000:1 | tee -a $O
  done
done
