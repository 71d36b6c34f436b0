#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
# the measurement switches below exist only in the -DCPI_EXPERIMENTS build (python -m cpi_amd.build --experiments)
export CPI_AMD_LIB=$PWD/cpi_amd/libcpi_amd_exp.so
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
D=/tmp/pmc_$$; mkdir -p $D; cd /tmp
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  n=$(echo $grp | cut -d" " -f1)
  timeout 300 rocprofv3 --pmc $grp -d $D -o pmc_$n -- python $R/tools/microbench.py "$@" > /dev/null 2> $D/err_$n.txt || tail -3 $D/err_$n.txt
done
python $R/tools/pmc_summary.py "$D/**/*.db" > $R/gpurun_out/exp_pmc_cov.txt
rm -rf $D
cat $R/gpurun_out/exp_pmc_cov.txt
