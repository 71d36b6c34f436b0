#!/bin/bash
# usage: tools/pmc_run.sh <lib.so> <out.txt> <microbench specs...>   (run on the GPU box; keeps only the text summary)
R=$PWD; LIB=$1; OUT=$2; shift 2
export TMPDIR=/tmp
D=/tmp/pmc_$$; mkdir -p $D; cd /tmp
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  n=$(echo $grp | cut -d" " -f1)
  CPI_AMD_LIB=$LIB timeout 300 rocprofv3 --pmc $grp -d $D -o pmc_$n -- python $R/tools/microbench.py "$@" > /dev/null 2> $D/err_$n.txt
done
python $R/tools/pmc_summary.py "$D/**/*.db" > $R/$OUT
rm -rf $D
cat $R/$OUT
