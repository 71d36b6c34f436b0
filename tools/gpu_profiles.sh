#!/bin/bash
# Round-2 measurement artefacts (run through gpurun): GPU tests, PMC counters of bench.py's rows (build-stamped JSON), the
# default bench command with the counters in place, and its kernel-trace summary.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r02_pytest_gpu.txt; cat gpurun_out/r02_pytest_gpu.txt
bash tools/pmc_collect.sh profiles/r02_pmc.json > gpurun_out/r02_pmc_collect.log 2>&1; tail -5 gpurun_out/r02_pmc_collect.log
cp profiles/r02_pmc.json gpurun_out/r02_pmc.json
python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err; tail -c 600 gpurun_out/r02_bench_default.json
bash tools/prof_run.sh gpurun_out/r02_kernel_stats.md --steps 20 --warmup 5 > /dev/null 2>&1; head -12 gpurun_out/r02_kernel_stats.md | cut -c1-160
CPI_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 20 --warmup 5 --workload v2_full --scaling strong --no-extra --no-cpu > gpurun_out/r02_bench_v2full_strong_dist1.json 2> gpurun_out/r02_bench_v2full_strong_dist1.err; tail -c 700 gpurun_out/r02_bench_v2full_strong_dist1.json
CPI_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 1 --steps 20 --warmup 5 --gather all --no-extra --no-cpu > gpurun_out/r02_bench_dist1_allgather.json 2> gpurun_out/r02_bench_dist1_allgather.err; tail -c 400 gpurun_out/r02_bench_dist1_allgather.json
python tools/profile_digest.py gpurun_out/r02_bench_default.json gpurun_out/r02_pmc.json > gpurun_out/r02_digest.md 2>/dev/null || true
