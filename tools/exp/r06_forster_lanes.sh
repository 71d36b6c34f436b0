#!/bin/bash
# Round 6 (late): the Forster comparator's mean and Jacobian-column steps on the six lanes of a window that carry them (default)
# against all sixteen (-DCPI_FORSTER_MEAN_LANES=16); steady state of long runs, alternating.
cd ${GRAFT_REPO_ROOT:-.}; R=$PWD
mkdir -p gpurun_out; O=$R/gpurun_out/r06_forster_lanes.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_forster.py tests/test_gpu_packed.py -x -q 2>&1 | tail -2 >> $O
ROWS="forster_full:100000:0:300 forster_full:1000000:0:30"
for round in 1 2 3; do
  for lib in libcpi_amd_fm16.so libcpi_amd.so; do
    CPI_AMD_LIB=$R/cpi_amd/$lib python tools/microbench.py $ROWS 2>&1 | grep launch_us >> $O
  done
done
cat $O
