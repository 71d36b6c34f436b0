#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
CPI_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --steps 5 --warmup 2 --no-extra --no-cpu > gpurun_out/r04_reh1.out 2> gpurun_out/r04_reh1.err; echo rc=$?; grep -v "^\[W\|^W0\|socket.cpp" gpurun_out/r04_reh1.err | head -60; tail -c 600 gpurun_out/r04_reh1.out
CPI_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --steps 5 --warmup 2 --no-extra --no-cpu --workload v2_full --windows 20000 --scaling strong > gpurun_out/r04_reh2.out 2> gpurun_out/r04_reh2.err; echo rc=$?; grep -v "^\[W\|^W0\|socket.cpp" gpurun_out/r04_reh2.err | head -60; tail -c 600 gpurun_out/r04_reh2.out
python -m pytest "tests/test_gpu_parity.py::test_config2_size_launch_geometries_vs_reference_sample" -m gpu -q 2>&1 | tail -30
