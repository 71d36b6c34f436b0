#!/bin/bash
# usage (on the GPU box, through gpurun): tools/pmc_collect.sh <out.json> [row ...]
# Collects the HBM-traffic and FP64-instruction counters of bench.py's rows with the IN-TREE library, one rocprofv3 --pmc
# pass per counter group (kernel-trace / stats domains are NOT combined with --pmc), and writes profiles-style JSON stamped
# with the library's build id.  A row is <workload>:<W>:<N>.
R=$PWD; OUT=$1; shift
ROWS=${@:-"v1_mean:10000:50 v1_mean:1000000:50 v1_full:100000:50 v2_full:100000:50 forster_full:100000:50 factor_v1:1000000:50 factor_v2:1000000:50 factor_v1_packed:1000000:50 factor_v2_packed:1000000:50 sqrt_info:1000000:50 factor_v1_whitened:1000000:50 factor_v2_whitened:1000000:50 factor_v1_hessian:1000000:50 factor_v2_hessian:1000000:50 predict_v1:1000000:50 predict_v2:1000000:50 cfg5_mean:1000000:100 cfg5_full:1000000:100 v1_mean_tiled:1000000:50 v2_mean_tiled:1000000:50 v1_mean_tiled:10000:50 v1_mean_stream:1000000:50 v1_full_stream:100000:50 v2_full_stream:100000:50"}
export TMPDIR=/tmp
D=/tmp/pmc_$$; mkdir -p $D; cd /tmp
BID=$(python -c "import sys; sys.path.insert(0,'$R'); from cpi_amd import _lib; print(_lib.load().cpi_build_id().decode())")
SPECS=()
for row in $ROWS; do
  wl=${row%%:*}; rest=${row#*:}; W=${rest%%:*}; N=${rest#*:}
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVES" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    CPI_MB_EAGER=1 CPI_MB_SAMPLES=$N timeout 300 rocprofv3 --pmc $grp -d $D/${wl}_${W}_${N} -o g$i -- python $R/tools/microbench.py $wl:$W:0:3 > /dev/null 2> $D/err_${wl}_$i.txt || tail -3 $D/err_${wl}_$i.txt
  done
  case $wl in
    v1_mean|cfg5_mean) K="cpi_mean_kernel<1, false";;
    v2_mean) K="cpi_mean_kernel<2, false";;
    v1_mean_tiled) K="cpi_mean_tiled_kernel<1";;
    v1_mean_stream) K="cpi_mean_kernel<1, false;cpi_cut_windows_kernel";;
    v1_full_stream) K="cpi_cov_kernel<1;cpi_mean_kernel<1, true;cpi_cut_windows_kernel";;
    v2_full_stream) K="cpi_cov_kernel<2;cpi_cut_windows_kernel";;
    v2_mean_tiled) K="cpi_mean_tiled_kernel<2";;
    sqrt_info|sqrt_info_packed) K="cpi_sqrt_info_kernel";;
    factor_v1_whitened|factor_v1_whitened_tri) K="cpi_factor_kernel<1, true";;
    factor_v2_whitened|factor_v2_whitened_tri) K="cpi_factor_kernel<2, true";;
    factor_v1_hessian|factor_v1_hessian_tri) K="cpi_factor_hessian_kernel<1";;
    factor_v2_hessian|factor_v2_hessian_tri) K="cpi_factor_hessian_kernel<2";;
    predict_v1) K="cpi_predict_kernel<1";;
    predict_v2) K="cpi_predict_kernel<2";;
    v1_full|cfg5_full|v1_full_sym|cfg5_full_sym) K="cpi_cov_kernel<1;cpi_mean_kernel<1, true";;
    v2_full|v2_full_sym) K="cpi_cov_kernel<2";;
    forster_full) K="cpi_forster_kernel";;
    factor_v1_packed|factor_v2_packed) K="cpi_factor_packed_kernel";;
    factor_v1) K="cpi_factor_kernel<1, false";;
    factor_v2) K="cpi_factor_kernel<2, false";;
    *) K="cpi_";;
  esac
  SPECS+=("$row|$K=$D/${wl}_${W}_${N}/**/*.db")
done
python $R/tools/pmc_summary.py --json $BID $R/$OUT "${SPECS[@]}"
rm -rf $D
