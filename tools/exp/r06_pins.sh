#!/bin/bash
# Round 6: the evaluateError core with every step's results PINNED before the next step's loads are formed (-DCPI_CORE_PIN=1: 188 -> 90-96
# registers on the dense / packed sweeps, 194 -> 144 on the whitened one, 190 / 238 -> 128 on the Hessian one) and the whitened sweep at
# three wavefronts per SIMD with R parked in the H stage (-DCPI_FACTOR_W3=1).  Parity of the variants, then alternating launch times.
# usage: tools/exp/r06_pins.sh   (needs cpi_amd/libcpi_amd_{pin,w3p,w3pe}.so: python -m cpi_amd.build --custom ...)
cd ${GRAFT_REPO_ROOT:-.}; R=$PWD
mkdir -p gpurun_out
O=$R/gpurun_out/r06_pins.txt
: > $O
for t in pin w3p; do
  echo "== parity, lib=$t" >> $O
  CPI_AMD_LIB=$R/cpi_amd/libcpi_amd_$t.so timeout 1200 python -m pytest tests/test_gpu_whitening.py tests/test_gpu_packed.py tests/test_gpu_parity.py -x -q -k "whiten or hessian or factor or sqrt or packed" 2>&1 | tail -3 >> $O
done
mb() { local lib=cpi_amd/libcpi_amd_$1.so; [ $1 = default ] && lib=cpi_amd/libcpi_amd.so; CPI_AMD_LIB=$R/$lib python tools/microbench.py "${@:2}" 2>&1 | grep -E "launch_us|rror" | sed "s/^/$1 /"; }
for round in 1 2; do for t in default pin w3p; do
  mb $t factor_v1:1000000:0 factor_v2:1000000:0 factor_v1_packed:1000000:0 factor_v2_packed:1000000:0 factor_v1_whitened:1000000:0 factor_v1_whitened_tri:1000000:0 \
      factor_v2_whitened:1000000:0 factor_v2_whitened_tri:1000000:0 factor_v1_hessian:1000000:0 factor_v1_hessian_tri:1000000:0 \
      factor_v2_hessian:1000000:0 factor_v2_hessian_tri:1000000:0 factor_v1:100000:0:100 factor_v1_packed:100000:0:100 >> $O
done; done
echo "== lanes per factor (experiments build with the pins)" >> $O
for lpf in 3 4 6 8; do CPI_AMD_PACKED_LPF=$lpf mb w3pe factor_v1_packed:1000000:0 factor_v2_packed:1000000:0 | sed "s/^/packed_lpf=$lpf /" >> $O; done
for l in 4 8 16; do CPI_AMD_FACTOR_LANES=$l mb w3pe factor_v1:1000000:0 factor_v2:1000000:0 factor_v1_whitened_tri:1000000:0 | sed "s/^/factor_lanes=$l /" >> $O; done
cat $O
