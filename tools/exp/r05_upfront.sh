#!/bin/bash
# Round 5: the small-batch mean kernel with ALL chunks of a lane-segment requested at once (-DCPI_MEAN_UPFRONT=1) against one chunk ahead.
# usage (GPU box): tools/exp/r05_upfront.sh <tag> ...   ("default" = shipped)
cd ${GRAFT_REPO_ROOT:-.}; R=$PWD
mkdir -p gpurun_out
O=gpurun_out/r05_upfront.txt
: > $O
for t in "$@"; do
  [ $t = default ] && continue
  echo "== parity, lib=$t" >> $O
  CPI_AMD_LIB=$R/cpi_amd/libcpi_amd_$t.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "lane_splits or config2 or fuzz or edge_sizes or seeded or segment_form" 2>&1 | tail -2 >> $O
done
mb() { local lib=cpi_amd/libcpi_amd_$1.so; [ $1 = default ] && lib=cpi_amd/libcpi_amd.so; CPI_AMD_LIB=$R/$lib python tools/microbench.py "${@:2}" 2>&1 | grep "launch_us" | sed "s/^/$1 /"; }
for round in 1 2 3; do for t in "$@"; do
  mb "$t" v1_mean:10000:0:2000 v1_mean:8000:0:2000 v1_mean:12000:0:2000 v1_mean:5000:0:2000 v2_mean:10000:0:1000 v1_mean:10000:5:1000 v1_mean:10000:8:1000 >> $O
done; done
cat $O
