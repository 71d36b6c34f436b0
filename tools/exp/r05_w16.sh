#!/bin/bash
# Round 5: 16-byte staged loads on the two-knot mean kernel's fast path (-DCPI_MEAN_W16=1), with and without the three-knot BIG kernel.
# usage (GPU box): tools/exp/r05_w16.sh <tag> ...   ("default" = shipped)
cd ${GRAFT_REPO_ROOT:-.}; R=$PWD
mkdir -p gpurun_out
O=gpurun_out/r05_w16.txt
: > $O
for t in "$@"; do
  [ $t = default ] && continue
  echo "== parity, lib=$t" >> $O
  CPI_AMD_LIB=$R/cpi_amd/libcpi_amd_$t.so timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_stream.py tests/test_gpu_full_size.py -x -q -m gpu -k "three_knot or lane_splits or config2 or fuzz or edge_sizes or seeded or segment_form or stream_entry or config5_one_gpu_share_1M_windows_x_100 or golden" 2>&1 | tail -2 >> $O
done
mb() { local lib=cpi_amd/libcpi_amd_$1.so; [ $1 = default ] && lib=cpi_amd/libcpi_amd.so; CPI_AMD_LIB=$R/$lib python tools/microbench.py "${@:2}" 2>&1 | grep "launch_us" | sed "s/^/$1 /"; }
for round in 1 2; do for t in "$@"; do
  mb "$t" v1_mean:1000000:1 v2_mean:1000000:1 v1_mean:300000:1 v1_mean:100000:1 v1_mean:30000:0:500 v1_mean:10000:0:2000 v1_mean_stream:1000000:0 v1_mean_stream:100000:0 >> $O
  CPI_MB_SAMPLES=100 mb "$t" v1_mean:1000000:1 | sed "s/$/  (100 samples)/" >> $O
done; done
cat $O
