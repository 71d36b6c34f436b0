#!/bin/bash
# usage (GPU box): tools/exp/pmc_fetch.sh <out.txt> <kernel filter> <microbench row> <lib> [<lib> ...]
# HBM traffic of one kernel under several builds: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (no trace domains);
# traffic = (2 * FETCH_SIZE + WRITE_SIZE) KiB (MI355X_MICROARCH.md "HBM").
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD; OUT=$1; FILT=$2; ROW=$3; shift 3
mkdir -p gpurun_out
export TMPDIR=/tmp
: > $R/$OUT
for LIB in "$@"; do
  D=/tmp/pmcf_$$_$(basename $LIB .so); mkdir -p $D; cd /tmp
  for grp in FETCH_SIZE WRITE_SIZE; do
    CPI_AMD_LIB=$R/$LIB CPI_MB_EAGER=1 timeout 300 rocprofv3 --pmc $grp -d $D -o pmc_$grp -- python $R/tools/microbench.py $ROW > $D/out_$grp.txt 2> $D/err_$grp.txt || tail -3 $D/err_$grp.txt
  done
  echo "=== $LIB  ($ROW)  $(grep launch_us $D/out_FETCH_SIZE.txt | head -1)" >> $R/$OUT
  python $R/tools/pmc_summary.py "$D/**/*.db" | grep "$FILT" | sed "s/^[^ ]* *//" >> $R/$OUT
  rm -rf $D
done
cat $R/$OUT
