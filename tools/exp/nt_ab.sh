#!/bin/bash
# A/B of the non-temporal hint on streaming stores: the assembler's tile rows (CPI_ASM_NT bit 1), the dense sweep's H1 / H2
# (CPI_FACTOR_NT bit 1), square-root information (bit 2), Hessian blocks (bit 3).
cd ${GRAFT_REPO_ROOT:-.}
mb() { CPI_AMD_LIB=$PWD/cpi_amd/libcpi_amd$1.so python tools/microbench.py "${@:2}" 2>&1 | grep "launch_us\|assembly"; }
for v in "" _ant2 "" _ant2; do mb "$v" v1_mean_tiled:1000000:0 | grep assembly; done
for v in "" _fnt2 _fnt6 _fnt10 _fnt14 "" _fnt2 _fnt6 _fnt10 _fnt14; do mb "$v" factor_v1:1000000:0 sqrt_info:1000000:0 factor_v1_hessian:1000000:0 factor_v2_hessian:1000000:0 factor_v1_whitened:1000000:0; done
