#!/bin/bash
# BIG mean kernel: wavefronts per CU set by unused dynamic LDS (-DCPI_MEAN_BIG_LDS_PAD): time and FETCH_SIZE of the dense and the
# stream-entry 1 M launches at 8 / 6 / 5 / 4 wavefronts per CU.
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
mkdir -p gpurun_out
O=$R/gpurun_out/r04_big_occ.txt
: > $O
export TMPDIR=/tmp
mb() { local lib=cpi_amd/libcpi_amd${1:+_$1}.so; CPI_AMD_LIB=$R/$lib python tools/microbench.py "${@:2}" 2>&1 | grep "launch_us"; }
for round in 1 2; do for t in "" occ6 occ5 occ4; do mb "$t" v1_mean:1000000:0 v1_mean_stream:1000000:0 v1_mean:200000:0 v2_mean:1000000:0 | tee -a $O; done; done
for t in "" occ4; do lib=cpi_amd/libcpi_amd${t:+_$t}.so; CPI_AMD_LIB=$R/$lib python tools/exp/stream_irregular.py 1000000 50 1 2>&1 | grep "model 1" | tee -a $O; done
cd /tmp
for t in "" occ6 occ5 occ4; do
  lib=cpi_amd/libcpi_amd${t:+_$t}.so
  for row in v1_mean:1000000:0:3 v1_mean_stream:1000000:0:3; do
    D=/tmp/po_$$_${t}_${row%%:*}; mkdir -p $D
    CPI_AMD_LIB=$R/$lib CPI_MB_EAGER=1 timeout 200 rocprofv3 --pmc FETCH_SIZE -d $D -o f -- python $R/tools/microbench.py $row > /dev/null 2> $D/err.txt || tail -3 $D/err.txt
    echo "=== FETCH_SIZE $lib $row" | tee -a $O
    python $R/tools/pmc_summary.py "$D/**/*.db" | grep "cpi_mean_kernel" | sed "s/^[^ ]* *//" | tee -a $O
    rm -rf $D
  done
done
