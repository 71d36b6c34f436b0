#!/bin/bash
# One GPU session: tests, default bench, dist path with one rank, cfg5 rows, counter list.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r02_pytest_gpu.txt; cat gpurun_out/r02_pytest_gpu.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err; tail -c 3000 gpurun_out/r02_bench_default.json; tail -5 gpurun_out/r02_bench_default.err
CPI_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu > gpurun_out/r02_bench_dist1.json 2> gpurun_out/r02_bench_dist1.err; tail -c 1500 gpurun_out/r02_bench_dist1.json; tail -3 gpurun_out/r02_bench_dist1.err
CPI_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29532 bench.py --gpus 1 --steps 5 --warmup 2 --workload cfg5_full --no-extra --no-cpu > gpurun_out/r02_bench_cfg5full_dist1.json 2> gpurun_out/r02_bench_cfg5full_dist1.err; tail -c 1500 gpurun_out/r02_bench_cfg5full_dist1.json; tail -3 gpurun_out/r02_bench_cfg5full_dist1.err
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -i -o "SQ_INSTS_VALU[A-Z0-9_]*\|FETCH_SIZE\|WRITE_SIZE\|SQ_INSTS_[A-Z0-9_]*F64[A-Z0-9_]*" | sort -u | tr '\n' ' ') > gpurun_out/r02_counters.txt; cat gpurun_out/r02_counters.txt
