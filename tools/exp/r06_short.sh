#!/bin/bash
# Round 6, VERDICT r05 item 3: short windows (N = 10 / 20: the reference's imurate / camrate).  (a) knots per chunk of the one-lane
# streaming kernel at 1 M windows -- BIG with C = 3 (shipped), 2, 5 and the two-knot kernel without BIG; (b) lanes per window for small
# batches against the automatic choice; (c) cpi_sqrt_info_kernel with 2 / 4 wavefronts per workgroup.
cd ${GRAFT_REPO_ROOT:-.}; R=$PWD
mkdir -p gpurun_out
O=$R/gpurun_out/r06_short.txt
: > $O
mb() { local lib=cpi_amd/libcpi_amd_$1.so; [ $1 = default ] && lib=cpi_amd/libcpi_amd.so; CPI_AMD_LIB=$R/$lib python tools/microbench.py "${@:2}" 2>&1 | grep -E "launch_us|rror" | sed "s/^/$1 /"; }
echo "== (a) knots per chunk, 1 M windows" >> $O
for N in 10 20; do for round in 1 2; do for t in default bigc2 bigc5 nobig; do
  CPI_MB_SAMPLES=$N mb $t v1_mean:1000000:0:40 v2_mean:1000000:0:40 v1_mean_stream:1000000:0:40 | sed "s/^/N=$N /" >> $O
done; done; done
echo "== (b) lanes per window, small batches" >> $O
for N in 10 20; do
  for W in 5000 10000 20000 30000 50000; do
    CPI_MB_SAMPLES=$N mb default v1_mean:$W:0:800 v1_mean:$W:1:800 v1_mean:$W:2:800 v1_mean:$W:3:800 v1_mean:$W:4:800 v1_mean:$W:5:800 v1_mean:$W:6:800 v1_mean:$W:8:800 | sed "s/^/N=$N /" >> $O
  done
  CPI_MB_SAMPLES=$N mb default v2_mean:10000:0:800 v2_mean:10000:1:800 v2_mean:10000:2:800 v2_mean:10000:3:800 v2_mean:10000:4:800 v2_mean:10000:5:800 v2_mean:10000:6:800 | sed "s/^/N=$N /" >> $O
done
echo "== (c) sqrt-information kernel: wavefronts per workgroup" >> $O
for round in 1 2; do for t in default sq2 sq4; do mb $t sqrt_info:1000000:0 sqrt_info_packed:1000000:0 >> $O; done; done
cat $O
