#!/bin/bash
# Round 6: the Hessian sweep with ONE time-shared exchange array and a two-factor output stage (11.1 / 14.5 KB of LDS, three wavefronts per SIMD).
cd ${GRAFT_REPO_ROOT:-.}; R=$PWD
mkdir -p gpurun_out
O=$R/gpurun_out/r06_hessian.txt
: > $O
timeout 1200 python -m pytest tests/test_gpu_whitening.py tests/test_gpu_packed.py tests/test_gpu_parity.py -x -q -k "hessian or packed or whiten" 2>&1 | tail -4 >> $O
for round in 1 2 3; do
  python tools/microbench.py factor_v1_hessian:1000000:0 factor_v1_hessian_tri:1000000:0 factor_v2_hessian:1000000:0 factor_v2_hessian_tri:1000000:0 factor_v1_whitened_tri:1000000:0 factor_v1_hessian_tri:100000:0:100 2>&1 | grep -E "launch_us|rror" >> $O
done
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -n 1 | wc -c >> $O
cat $O
