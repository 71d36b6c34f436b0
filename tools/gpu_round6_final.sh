#!/bin/bash
# Round-6 measurement session (run through gpurun): PMC counters of the key bench rows (build-stamped JSON) incl. the packed-triangle
# rows of ABI 3, the rebuilt small sweeps and the short-window rows; the driver's bench command with the counters in place; kernel-trace
# summaries (driver command, headline only, one process per row); the multi-GPU rehearsals (N = 2, 4, 8 on one GPU through gloo) with the
# `predicted` wire budget in the line, and the exchange INSIDE a batch (chunked schedule, 1 vs 8 sub-blocks) on the full-V1 share of
# configs[4].  Everything lands in gpurun_out/ (r06_*); the files worth keeping are copied to profiles/.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
PMC_ROWS="v1_mean:10000:50 v1_mean:1000000:50 v1_full:100000:50 v2_full:100000:50 factor_v1:1000000:50 factor_v2:1000000:50 factor_v1_packed:1000000:50 factor_v2_packed:1000000:50 sqrt_info:1000000:50 sqrt_info_packed:1000000:50 factor_v1_whitened:1000000:50 factor_v1_whitened_tri:1000000:50 factor_v1_hessian:1000000:50 factor_v1_hessian_tri:1000000:50 predict_v1:1000000:50 predict_v2:1000000:50 cfg5_mean:1000000:100 v1_mean_tiled:1000000:50 v1_mean_stream:1000000:50 v1_mean:1000000:10 v1_mean:1000000:20 v1_mean:10000:10"
bash tools/pmc_collect.sh profiles/r06_pmc.json $PMC_ROWS > gpurun_out/r06_pmc_collect.log 2>&1; tail -3 gpurun_out/r06_pmc_collect.log
cp profiles/r06_pmc.json gpurun_out/r06_pmc.json
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_default.out 2> gpurun_out/r06_bench_default.err; tail -n 1 gpurun_out/r06_bench_default.out > gpurun_out/r06_bench_default.json; wc -c gpurun_out/r06_bench_default.json; cp bench_extra.json gpurun_out/r06_bench_extra.json
for wl in v2_full factor_v1 cfg5_mean cfg5_full; do python bench.py --workload $wl --steps 20 --warmup 3 --no-extra 2>/dev/null | tail -n 1 > gpurun_out/r06_bench_$wl.json; done
bash tools/prof_run.sh gpurun_out/r06_kernel_stats.md --gpus 1 --steps 20 --warmup 5 > /dev/null 2>&1; head -8 gpurun_out/r06_kernel_stats.md | cut -c1-150
bash tools/prof_run.sh gpurun_out/r06_kernel_stats_headline.md --steps 2000 --warmup 200 --no-extra > /dev/null 2>&1; head -5 gpurun_out/r06_kernel_stats_headline.md | cut -c1-150
bash tools/row_traces.sh gpurun_out/r06_kernel_stats_rows.md > /dev/null 2>&1; grep -c "^###" gpurun_out/r06_kernel_stats_rows.md
CPI_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu 2>/dev/null | tail -n 1 > gpurun_out/r06_bench_dist1_rccl.json; tail -c 300 gpurun_out/r06_bench_dist1_rccl.json; echo
for n in 2 4 8; do
  CPI_BENCH_SINGLE_DEVICE=1 python bench.py --gpus $n --steps 20 --warmup 5 2>/dev/null | tail -n 1 > gpurun_out/r06_bench_gpus${n}_rehearsal.json; tail -c 200 gpurun_out/r06_bench_gpus${n}_rehearsal.json; echo
done
CPI_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 8 --steps 5 --warmup 2 --workload cfg5_mean 2>/dev/null | tail -n 1 > gpurun_out/r06_bench_gpus8_cfg5_mean_rehearsal.json; tail -c 300 gpurun_out/r06_bench_gpus8_cfg5_mean_rehearsal.json; echo
# the exchange inside one batch, full-V1 share of configs[4] per rank (1 M windows x 100 samples), dense P / packed P, 1 / 8 sub-blocks
for spec in "cfg5_full 1" "cfg5_full 8" "cfg5_full_sym 8"; do set -- $spec
  CPI_BENCH_SINGLE_DEVICE=1 CPI_BENCH_STRICT=1 timeout 1500 python bench.py --gpus 8 --steps 2 --warmup 1 --workload $1 --gather-schedule chunked --gather-chunks $2 2> gpurun_out/r06_chunked_$1_k$2.err | tail -n 1 > gpurun_out/r06_bench_gpus8_$1_chunked_k$2_rehearsal.json
  tail -c 400 gpurun_out/r06_bench_gpus8_$1_chunked_k$2_rehearsal.json; echo; tail -2 gpurun_out/r06_chunked_$1_k$2.err
done
python tools/profile_digest.py gpurun_out/r06_bench_extra.json gpurun_out/r06_pmc.json > gpurun_out/r06_digest.md 2>/dev/null || true
