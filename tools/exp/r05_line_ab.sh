#!/bin/bash
# Round 5, VERDICT item 1(b): cpi_mean_line_kernel (whole 128-byte lines, phase-sorted wavefronts) against the shipped staged
# kernels.  usage (GPU box): tools/exp/r05_line_ab.sh <variant tag> [<variant tag> ...]   (cpi_amd/libcpi_amd_<tag>.so; "" = default)
# (The kernel now lives in the experiments build: python -m cpi_amd.build --custom <tag> -DCPI_EXPERIMENTS [...], run with CPI_AMD_MEAN_LINE=1;
# the session recorded in profiles/r05_mean_traffic.md used an earlier -DCPI_MEAN_LINE_W=<windows> switch of the default build.)
# (i) bitwise check of the variant against the two-knot kernel + reference sample (the three-knot test runs whatever kernel the
# library picks for a one-lane dense launch), (ii) same-box alternating launch times, (iii) FETCH_SIZE / WRITE_SIZE per launch.
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$R/gpurun_out/r05_line_ab.txt
: > $O
TAGS="$@"
for t in $TAGS; do
  echo "== parity, lib=$t" >> $O
  CPI_AMD_LIB=$R/cpi_amd/libcpi_amd_$t.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "three_knot or config5_one_gpu_share_1M_windows_x_100 or fuzz" 2>&1 | tail -3 >> $O
done
mb() { local lib=cpi_amd/libcpi_amd${1:+_$1}.so; CPI_AMD_LIB=$R/$lib python tools/microbench.py "${@:2}" 2>&1 | grep "launch_us" | sed "s/^/${1:-default} /"; }
for round in 1 2; do for t in "" $TAGS; do
  mb "$t" v1_mean:1000000:1 v2_mean:1000000:1 v1_mean:300000:1 v1_mean:100000:1 >> $O
  CPI_MB_SAMPLES=100 mb "$t" v1_mean:1000000:1 | sed "s/$/  (100 samples)/" >> $O
done; done
cd /tmp
for t in "" $TAGS; do
  lib=cpi_amd/libcpi_amd${t:+_$t}.so
  for grp in FETCH_SIZE WRITE_SIZE; do
    D=/tmp/r05l_$$; rm -rf $D; mkdir -p $D
    CPI_AMD_LIB=$R/$lib CPI_MB_EAGER=1 timeout 200 rocprofv3 --pmc $grp -d $D -o f -- python $R/tools/microbench.py v1_mean:1000000:1:3 > /dev/null 2> $D/err.txt || tail -3 $D/err.txt >> $O
    echo "=== $grp lib=${t:-default} v1_mean 1 M x 50" >> $O
    python $R/tools/pmc_summary.py "$D/**/*.db" | grep "cpi_mean" | sed "s/^[^ ]* *//" >> $O
    rm -rf $D
  done
done
cat $O
