#!/bin/bash
# Round 5, VERDICT item 5: the assembler's wavefront-internal pipeline (-DCPI_ASM_PIPE=1) against the shipped kernel.
# usage (GPU box): tools/exp/r05_assembler.sh <tag> ...   (cpi_amd/libcpi_amd_<tag>.so; "default" = shipped)
cd ${GRAFT_REPO_ROOT:-.}; R=$PWD
mkdir -p gpurun_out
O=gpurun_out/r05_assembler.txt
: > $O
for t in "$@"; do
  [ $t = default ] && continue
  echo "== parity, lib=$t" >> $O
  CPI_AMD_LIB=$R/cpi_amd/libcpi_amd_$t.so timeout 900 python -m pytest tests/test_stream.py tests/test_gpu_tiled.py -x -q -m gpu -k "assembl or tiled" 2>&1 | tail -2 >> $O
done
mb() { local lib=cpi_amd/libcpi_amd_$1.so; [ $1 = default ] && lib=cpi_amd/libcpi_amd.so; CPI_AMD_LIB=$R/$lib python tools/microbench.py "${@:2}" 2>&1 | grep -E "assembly" | sed "s/^/$1 /"; }
for round in 1 2 3; do for t in "$@"; do mb "$t" v1_mean_tiled:1000000:0 v1_mean_tiled:100000:0 >> $O; done; done
libs=""; for t in "$@"; do if [ $t = default ]; then libs="$libs cpi_amd/libcpi_amd.so"; else libs="$libs cpi_amd/libcpi_amd_$t.so"; fi; done
bash tools/exp/pmc_fetch.sh gpurun_out/r05_assembler_pmc.txt cpi_assemble v1_mean_tiled:1000000:0:3 $libs > /dev/null 2>&1
cat gpurun_out/r05_assembler_pmc.txt >> $O
cat $O
