#!/bin/bash
# HBM traffic of the three mean-kernel designs at 1 M x 50 (calibration: the block-resident kernel reads every byte once, linearly)
cd ${GRAFT_REPO_ROOT:-.}
# the measurement switches below exist only in the -DCPI_EXPERIMENTS build (python -m cpi_amd.build --experiments)
export CPI_AMD_LIB=$PWD/cpi_amd/libcpi_amd_exp.so
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$R/gpurun_out/exp_pmc_mean.txt
: > $OUT
for cfg in "old" "blk8" "dma420"; do
  D=/tmp/pmcm_$cfg; mkdir -p $D; cd /tmp
  unset CPI_AMD_MEAN_BLK CPI_AMD_MEAN_DMA
  [ $cfg = blk8 ] && export CPI_AMD_MEAN_BLK=8
  [ $cfg = dma420 ] && export CPI_AMD_MEAN_DMA=4,2,0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
    n=$(echo $grp | cut -d" " -f1)
    timeout 300 rocprofv3 --pmc $grp -d $D -o p_$n -- python $R/tools/microbench.py v1_mean:1000000:1:3 > /dev/null 2> $D/err_$n.txt || tail -2 $D/err_$n.txt
  done
  echo "=== $cfg" >> $OUT
  python $R/tools/pmc_summary.py "$D/**/*.db" >> $OUT
  rm -rf $D
done
cat $OUT
