#!/bin/bash
# A/B of the non-temporal hint, second set: loads of R in the whitened (1) / Hessian (2) sweeps, of P in sqrt-information (4), packed stores (8)
cd ${GRAFT_REPO_ROOT:-.}
mb() { CPI_AMD_LIB=$PWD/cpi_amd/libcpi_amd$1.so python tools/microbench.py "${@:2}" 2>&1 | grep "launch_us"; }
for v in "" _ntl1 _ntl2 _ntl4 _ntl8 _ntl15 "" _ntl1 _ntl2 _ntl4 _ntl8 _ntl15; do mb "$v" factor_v1_whitened:1000000:0 factor_v1_hessian:1000000:0 sqrt_info:1000000:0 factor_v1_packed:1000000:0 factor_v2_packed:1000000:0; done
