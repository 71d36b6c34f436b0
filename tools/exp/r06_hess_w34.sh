cd ${GRAFT_REPO_ROOT:-.}; R=$PWD
mkdir -p gpurun_out; O=$R/gpurun_out/r06_hess_w34.txt; : > $O
ROWS="factor_v1_hessian_tri:1000000:0:300 factor_v2_hessian_tri:1000000:0:300 factor_v1_hessian_tri:100000:0:1000 factor_v1_hessian_tri:20000:0:2000"
for round in 1 2 3; do
  for lib in libcpi_amd_w3.so libcpi_amd.so; do
    CPI_AMD_LIB=$R/cpi_amd/$lib python tools/microbench.py $ROWS 2>&1 | grep launch_us >> $O
  done
done
cat $O
