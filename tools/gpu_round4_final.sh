#!/bin/bash
# Round-4 measurement session (run through gpurun): GPU tests, PMC counters of every bench row (build-stamped JSON), the driver's
# bench command with the counters in place, kernel-trace summaries (driver command, headline only, one process per row), the
# distributed paths that one GPU allows.  Everything lands in gpurun_out/ (r04_*); the files worth keeping are copied to profiles/.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r04_pytest_gpu.txt; cat gpurun_out/r04_pytest_gpu.txt
bash tools/pmc_collect.sh profiles/r04_pmc.json > gpurun_out/r04_pmc_collect.log 2>&1; tail -3 gpurun_out/r04_pmc_collect.log
cp profiles/r04_pmc.json gpurun_out/r04_pmc.json
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_default.out 2> gpurun_out/r04_bench_default.err; tail -n 1 gpurun_out/r04_bench_default.out > gpurun_out/r04_bench_default.json; wc -c gpurun_out/r04_bench_default.json; cp bench_extra.json gpurun_out/r04_bench_extra.json
for wl in v2_full factor_v1 cfg5_mean cfg5_full; do python bench.py --workload $wl --steps 20 --warmup 3 --no-extra 2>/dev/null | tail -n 1 > gpurun_out/r04_bench_$wl.json; done
bash tools/prof_run.sh gpurun_out/r04_kernel_stats.md --gpus 1 --steps 20 --warmup 5 > /dev/null 2>&1; head -8 gpurun_out/r04_kernel_stats.md | cut -c1-150
bash tools/prof_run.sh gpurun_out/r04_kernel_stats_headline.md --steps 2000 --warmup 200 --no-extra > /dev/null 2>&1; head -5 gpurun_out/r04_kernel_stats_headline.md | cut -c1-150
bash tools/prof_run.sh gpurun_out/r04_kernel_stats_headline_eager.md --steps 2000 --warmup 200 --no-extra --eager > /dev/null 2>&1; head -4 gpurun_out/r04_kernel_stats_headline_eager.md | cut -c1-150
bash tools/row_traces.sh gpurun_out/r04_kernel_stats_rows.md > /dev/null 2>&1; grep -c "^###" gpurun_out/r04_kernel_stats_rows.md
CPI_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu 2>/dev/null | tail -n 1 > gpurun_out/r04_bench_dist1_rccl.json; tail -c 300 gpurun_out/r04_bench_dist1_rccl.json; echo
CPI_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 1 --steps 20 --warmup 5 --workload v2_full --scaling strong --no-extra --no-cpu 2>/dev/null | tail -n 1 > gpurun_out/r04_bench_v2full_strong_1rank_rccl.json; tail -c 300 gpurun_out/r04_bench_v2full_strong_1rank_rccl.json; echo
CPI_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --steps 20 --warmup 5 --no-extra --no-cpu 2>/dev/null | tail -n 1 > gpurun_out/r04_bench_gpus2_rehearsal.json; tail -c 300 gpurun_out/r04_bench_gpus2_rehearsal.json; echo
python tools/profile_digest.py gpurun_out/r04_bench_extra.json gpurun_out/r04_pmc.json > gpurun_out/r04_digest.md 2>/dev/null || true
bash tests/tools/sanitize.sh gpu > /dev/null 2>&1; tail -12 gpurun_out/sanitize_gpu.txt
