#!/bin/bash
# usage (GPU box): tools/exp/r05_line_modes.sh <tag> ...   -- launch times of the dense 1 M x 50 one-lane mean launch under several
# builds (fetch-only / arithmetic-only builds of cpi_mean_line_kernel: -DCPI_MEAN_LINE_MODE=1 / 2), two alternating rounds
cd ${GRAFT_REPO_ROOT:-.}; R=$PWD
mb() { local lib=cpi_amd/libcpi_amd_$1.so; [ $1 = default ] && lib=cpi_amd/libcpi_amd.so; CPI_AMD_LIB=$R/$lib python tools/microbench.py "${@:2}" 2>&1 | grep "launch_us" | sed "s/^/$1 /"; }
for round in 1 2; do for t in "$@"; do mb "$t" v1_mean:1000000:1; done; done
