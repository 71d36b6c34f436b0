#!/bin/bash
# Round 4: cpi_mean_kernel<..., BIG> rebuilt on 32-bit staging offsets (three knots per chunk at TWO wavefronts per SIMD).
# Correctness of every L = 1 mean-only launch through BIG (variant bigall), then the same-box A/B against the library before the change.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r04_big2_ab.txt
: > $O
echo "== parity, every one-lane mean-only launch through BIG (libcpi_amd_bigall.so)" | tee -a $O
CPI_AMD_LIB=$PWD/cpi_amd/libcpi_amd_bigall.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_stream.py tests/test_gpu_full_size.py -m gpu -q -x 2>&1 | tail -4 | tee -a $O
echo "== stream tests, default library" | tee -a $O
timeout 300 python -m pytest tests/test_stream.py -m gpu -q -x 2>&1 | tail -2 | tee -a $O
mb() { local lib=cpi_amd/libcpi_amd${1:+_$1}.so; CPI_AMD_LIB=$PWD/$lib python tools/microbench.py "${@:2}" 2>&1 | grep "launch_us"; }
for round in 1 2; do
  for t in r4base "" bigd; do
    mb "$t" v1_mean:100000:1 v1_mean:200000:1 v1_mean:500000:0 v1_mean:1000000:0 v1_mean_stream:1000000:0 v1_mean_stream:200000:1 | tee -a $O
    CPI_MB_SAMPLES=100 mb "$t" v1_mean:1000000:0 | sed 's/v1_mean /v1_mean(N=100) /' | tee -a $O
  done
  for t in r4base bigall; do mb "$t" v2_mean:1000000:0 v2_mean:200000:1 | tee -a $O; done
done
for t in r4base ""; do
  lib=cpi_amd/libcpi_amd${t:+_$t}.so
  CPI_AMD_LIB=$PWD/$lib python tools/exp/stream_irregular.py 1000000 50 1 2>&1 | grep "us per launch" | tee -a $O
  CPI_AMD_LIB=$PWD/$lib python tools/exp/stream_irregular.py 1000000 50 0 2>&1 | grep "us per launch" | tee -a $O
done
