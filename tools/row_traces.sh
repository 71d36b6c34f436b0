#!/bin/bash
# usage (on the GPU box, through gpurun): tools/row_traces.sh <out.md> [row ...]
# One `rocprofv3 --kernel-trace --stats` PROCESS PER BENCH ROW (a row is <workload>:<W>:<N>:<steps>), so that the per-kernel
# average of a row is not mixed with launches of the same kernel at other batch sizes: the clean per-row launch duration the
# HIP-event time of bench.py is to be held against.  Set-up kernels of a row (the preintegration that produces a factor
# row's measurements, torch's generators) appear in its table too; the row's own kernels are the ones launched `steps` x 4
# times (3 timed repetitions + warm-up).
R=$PWD; OUT=$1; shift
ROWS=${@:-"v1_mean:10000:50:2000 v1_mean:30000:50:500 v1_mean:100000:50:200 v1_mean:1000000:50:20 v1_full:100000:50:20 v2_full:100000:50:20 forster_full:100000:50:20 factor_v1:1000000:50:20 factor_v2:1000000:50:20 factor_v1_packed:1000000:50:20 factor_v2_packed:1000000:50:20 sqrt_info:1000000:50:10 factor_v1_whitened:1000000:50:10 factor_v2_whitened:1000000:50:10 factor_v1_hessian:1000000:50:10 factor_v2_hessian:1000000:50:10 predict_v1:1000000:50:20 predict_v2:1000000:50:20 cfg5_mean:1000000:100:10 cfg5_full:1000000:100:3 v1_mean_tiled:1000000:50:20 v2_mean_tiled:1000000:50:20 v1_mean_tiled:10000:50:1000 v1_mean_stream:1000000:50:20 v1_full_stream:100000:50:20 v2_full_stream:100000:50:20 sqrt_info_packed:1000000:50:10 factor_v1_whitened_tri:1000000:50:10 factor_v2_whitened_tri:1000000:50:10 factor_v1_hessian_tri:1000000:50:10 factor_v2_hessian_tri:1000000:50:10 v1_full_sym:100000:50:20 v2_full_sym:100000:50:20 v1_mean:1000000:10:20 v1_mean:1000000:20:20 v1_mean:10000:10:2000 v1_mean:10000:20:2000 v1_full:1000000:10:3 v2_full:1000000:20:3 v1_mean_tiled:1000000:10:20 v1_mean_stream:1000000:20:20"}
export TMPDIR=/tmp
: > $R/$OUT
for row in $ROWS; do
  IFS=: read wl W N steps <<< "$row"
  D=/tmp/rowtrace_$$_$wl_$W; mkdir -p $D; cd /tmp
  CPI_MB_EAGER=1 CPI_MB_SAMPLES=$N timeout 600 rocprofv3 --kernel-trace --stats -d $D -o kt -- python $R/tools/microbench.py $wl:$W:0:$steps > $D/out.txt 2> $D/err.txt || tail -3 $D/err.txt
  echo "### $wl, $W units x $N samples per launch ($steps launches per repetition)" >> $R/$OUT
  grep launch_us $D/out.txt | sed 's/^/    HIP events: /' >> $R/$OUT
  echo >> $R/$OUT
  python $R/tools/kernel_stats.py "$D/**/*.db" | grep -E "^\| kernel|^\|---|cpi_" >> $R/$OUT
  echo >> $R/$OUT
  rm -rf $D
done
cat $R/$OUT
