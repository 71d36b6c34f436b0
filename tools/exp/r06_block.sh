#!/bin/bash
# Round 6: cpi_mean_block_kernel (N <= 11: the wavefront's knots as one linear LDS block) -- parity, then the short-window rows
# against the library before it (cpi_amd/libcpi_amd_prev.so: -DCPI_MEAN_BLOCK_NMAX=0 never admits the block kernel).
cd ${GRAFT_REPO_ROOT:-.}; R=$PWD
mkdir -p gpurun_out
O=$R/gpurun_out/r06_block.txt
: > $O
timeout 1500 python -m pytest tests/test_gpu_short_windows.py tests/test_stream.py tests/test_gpu_parity.py -x -q -m gpu -k "block or short or stream or lane or edge or fuzz or golden" 2>&1 | tail -4 >> $O
mb() { local lib=cpi_amd/libcpi_amd_$1.so; [ $1 = default ] && lib=cpi_amd/libcpi_amd.so; CPI_AMD_LIB=$R/$lib python tools/microbench.py "${@:2}" 2>&1 | grep -E "launch_us|rror" | sed "s/^/$1 /"; }
for N in 10 5; do for round in 1 2; do for t in default prev; do
  CPI_MB_SAMPLES=$N mb $t v1_mean:1000000:0:40 v2_mean:1000000:0:40 v1_mean_stream:1000000:0:40 v1_mean:100000:0:200 v1_mean:60000:0:300 v1_mean:60000:1:300 v1_mean:30000:0:500 v1_mean:30000:1:500 | sed "s/^/N=$N /" >> $O
done; done; done
cat $O
