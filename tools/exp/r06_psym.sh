#!/bin/bash
# Round 6 (late): cpi_cov_kernel<1> with rows p of F X sent through the exchange like the others (9 rows, no masked row_shr:6 moves:
# -120 of ~407 vector instructions per interval, +12 LDS writes) against the shipped symmetry route -- the trade was made in round 3
# when the LDS pipe bounded the kernel; the counters now say the vector pipe is 81 % busy.
cd ${GRAFT_REPO_ROOT:-.}; R=$PWD
mkdir -p gpurun_out
O=$R/gpurun_out/r06_psym.txt
: > $O
CPI_AMD_LIB=$R/cpi_amd/libcpi_amd_psym0.so timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_packed.py tests/test_gpu_short_windows.py -x -q 2>&1 | tail -3 >> $O
ROWS="v1_full:100000:0 v1_full:30000:0 v1_full_sym:100000:0 v1_full_stream:100000:0"
for round in 1 2 3; do
  for lib in libcpi_amd.so libcpi_amd_psym0.so; do
    CPI_AMD_LIB=$R/cpi_amd/$lib python tools/microbench.py $ROWS 2>&1 | grep launch_us >> $O
    CPI_AMD_LIB=$R/cpi_amd/$lib CPI_MB_SAMPLES=10 python tools/microbench.py v1_full:1000000:0 2>&1 | grep launch_us | sed 's/$/  (x10)/' >> $O
  done
done
cat $O
