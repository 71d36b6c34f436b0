#!/bin/bash
# usage (GPU box): tools/exp/pmc_ab.sh <out.txt> <kernel name filter> <microbench row> <lib A> [<lib B> ...]
# SQ counter groups of one kernel under two (or more) builds of the library, one rocprofv3 --pmc pass per group and build
# (no trace domains), summarised per kernel by tools/pmc_summary.py -- the before / after table of a kernel change.
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD; OUT=$1; FILT=$2; ROW=$3; shift 3
mkdir -p gpurun_out
export TMPDIR=/tmp
: > $R/$OUT
for LIB in "$@"; do
  D=/tmp/pmcab_$$_$(basename $LIB .so); mkdir -p $D; cd /tmp
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA" \
             "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
             "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    n=$(echo $grp | cut -d" " -f1)
    CPI_AMD_LIB=$R/$LIB CPI_MB_EAGER=1 timeout 300 rocprofv3 --pmc $grp -d $D -o pmc_$n -- python $R/tools/microbench.py $ROW > /dev/null 2> $D/err_$n.txt || tail -3 $D/err_$n.txt
  done
  echo "=== $LIB  ($ROW)" >> $R/$OUT
  python $R/tools/pmc_summary.py "$D/**/*.db" | grep "$FILT" | awk '{print $2, $3, $4, $5}' >> $R/$OUT
  rm -rf $D
done
cat $R/$OUT
