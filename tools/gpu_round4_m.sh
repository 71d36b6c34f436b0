#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee gpurun_out/r04_pytest_final.txt
CPI_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu 2>/dev/null | tail -n 1 > gpurun_out/r04_bench_dist1_rccl.json; tail -c 400 gpurun_out/r04_bench_dist1_rccl.json; echo
CPI_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 1 --steps 20 --warmup 5 --workload v2_full --scaling strong --no-extra --no-cpu 2>/dev/null | tail -n 1 > gpurun_out/r04_bench_v2full_strong_1rank_rccl.json; tail -c 300 gpurun_out/r04_bench_v2full_strong_1rank_rccl.json; echo
CPI_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --steps 20 --warmup 5 --no-extra --no-cpu 2>/dev/null | tail -n 1 > gpurun_out/r04_bench_gpus2_rehearsal.json; tail -c 400 gpurun_out/r04_bench_gpus2_rehearsal.json; echo
