#!/bin/bash
# Round-4 session C: fused cut + peeled tail + DMA assembler: correctness (stream / tiled / parity-geometry / group) and timing.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python -m pytest tests/test_stream.py tests/test_gpu_tiled.py "tests/test_gpu_parity.py::test_config2_size_launch_geometries_vs_reference_sample" tests/test_gpu_group.py::test_bench_gpus_2_end_to_end_rehearsal_on_one_gpu tests/test_gpu_full_size.py -m gpu -q 2>&1 | tail -150 > gpurun_out/r04_pytest_c.txt; tail -5 gpurun_out/r04_pytest_c.txt
python tools/microbench.py v1_mean_stream:1000000:0 v1_mean_tiled:1000000:0 v2_mean_tiled:1000000:0 v1_mean_stream:100000:0 v1_full_stream:100000:0 v2_full_stream:100000:0 2>&1 | tee gpurun_out/r04_mb_c.txt
