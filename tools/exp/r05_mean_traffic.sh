#!/bin/bash
# Round 5, VERDICT item 1(a): where are the dense / stream mean kernels' extra fetches served from?
# (i) which L2 fabric-side counters this rocprofv3 has on gfx950 (rocprofv3 -L); (ii) those counters on the shipped 1 M x 50 dense
# launch, the 1 M x 51 stream launch and the tiled kernel (the 1.04 x control), at 8 and at 4 wavefronts per CU
# (cpi_amd/libcpi_amd_occ4.so: python tools/exp/build_mean_variants.py occ4="-DCPI_MEAN_BIG_LDS_PAD=29000"); (iii) the rate probe
# tools/exp/mall_probe.hip.  One rocprofv3 --pmc pass per counter group (no trace domains).  Output: gpurun_out/r05_mean_traffic.txt
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$R/gpurun_out/r05_mean_traffic.txt
: > $O
echo "== rocprofv3 -L (TCC fabric-side counters present)" >> $O
(cd /tmp && timeout 120 rocprofv3 -L 2>/dev/null | grep -o -E "TCC_(EA0?_[A-Z0-9_]+|HIT[A-Za-z_]*|MISS[A-Za-z_]*|REQ[A-Za-z_]*|READ[A-Za-z_]*|TAG_STALL[A-Za-z_]*)" | sort -u | tr '\n' ' ') >> $O 2>&1
echo >> $O
echo "== mall_probe" >> $O
timeout 120 tools/exp/bin/mall_probe >> $O 2>&1
ROWS="v1_mean:1000000:0:3 v1_mean_stream:1000000:0:3 v1_mean_tiled:1000000:0:3"
pass() {  # <lib tag> <counter group...>
  local t=$1; shift
  local lib=cpi_amd/libcpi_amd${t:+_$t}.so
  local D=/tmp/r05mt_$$; rm -rf $D; mkdir -p $D
  (cd /tmp && CPI_AMD_LIB=$R/$lib CPI_MB_EAGER=1 timeout 300 rocprofv3 --pmc "$@" -d $D -o p -- python $R/tools/microbench.py $ROWS > $D/out.txt 2> $D/err.txt) || tail -3 $D/err.txt >> $O
  echo "=== lib=${t:-default} pmc: $*" >> $O
  grep launch_us $D/out.txt | sed "s/^[^ ]* *//" >> $O
  python $R/tools/pmc_summary.py "$D/**/*.db" | grep -E "cpi_mean" | sed "s/^[^ ]* *//" >> $O
  rm -rf $D
}
for t in "" occ4; do
  pass "$t" FETCH_SIZE
  pass "$t" TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum
  pass "$t" TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_BUBBLE_sum
done
pass "" TCC_EA0_RDREQ_DRAM_32B TCC_EA0_RDREQ_GMI_32B TCC_EA0_RDREQ_IO_32B
pass "" TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B TCC_EA0_RD_UNCACHED_32B_sum
pass "" TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pass "" TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum
cat $O
