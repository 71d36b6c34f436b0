#!/bin/bash
# SQ counter groups of the dense 1 M x 50 one-lane mean launch under several builds (round 5: why the line kernel is not faster)
# usage (GPU box): tools/exp/r05_line_sq.sh <tag> [<tag> ...]   ("default" = the shipped library)
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$R/gpurun_out/r05_line_sq.txt
: > $O
for t in "$@"; do
  lib=cpi_amd/libcpi_amd_$t.so; [ $t = default ] && lib=cpi_amd/libcpi_amd.so
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR" \
             "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH" \
             "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    D=/tmp/r05sq_$$; rm -rf $D; mkdir -p $D
    (cd /tmp && CPI_AMD_LIB=$R/$lib CPI_MB_EAGER=1 timeout 200 rocprofv3 --pmc $grp -d $D -o p -- python $R/tools/microbench.py v1_mean:1000000:1:3 > $D/out.txt 2> $D/err.txt) || tail -3 $D/err.txt >> $O
    echo "=== lib=$t $(grep launch_us $D/out.txt | sed 's/.*launch_us= *//; s/ .*//') us" >> $O
    python $R/tools/pmc_summary.py "$D/**/*.db" | grep -E "cpi_mean_line|cpi_mean_kernel<1, false, false, 1, 0, true>" | sed "s/^[^ ]* *//" >> $O
    rm -rf $D
  done
done
cat $O
