#!/bin/bash
# Round-4 session A: the whole GPU suite on the new library (hooks split out, over-read fix), the driver's bench command
# (line < 6 KB + bench_extra.json), the 1-rank RCCL path with the new rccl / gather_verified fields.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r04_pytest_a.txt; cat gpurun_out/r04_pytest_a.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_default.out 2> gpurun_out/r04_bench_default.err; tail -c 1500 gpurun_out/r04_bench_default.err
tail -n 1 gpurun_out/r04_bench_default.out > gpurun_out/r04_bench_default.json
wc -c gpurun_out/r04_bench_default.json
cp bench_extra.json gpurun_out/r04_bench_extra.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_default.json').read())
print("HEADLINE", d["value"], d["ms_per_step"], d["roofline"]["launch_us"], d["roofline"]["frac"], d.get("goal_40pct_hbm"))
print("configs2", json.dumps(d.get("configs2"))[:600])
print("routes", d.get("routes_1M_x_50"))
for k, v in d.get("extra_rows", {}).items(): print("  %-32s %s" % (k, v))
PY
CPI_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu > gpurun_out/r04_bench_dist1.out 2> gpurun_out/r04_bench_dist1.err; tail -n 1 gpurun_out/r04_bench_dist1.out | tee gpurun_out/r04_bench_dist1_rccl.json | tail -c 2500; tail -3 gpurun_out/r04_bench_dist1.err
CPI_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29532 bench.py --gpus 1 --steps 20 --warmup 5 --workload v2_full --scaling strong --no-extra --no-cpu > gpurun_out/r04_bench_v2full_dist1.out 2> gpurun_out/r04_bench_v2full_dist1.err; tail -n 1 gpurun_out/r04_bench_v2full_dist1.out | tee gpurun_out/r04_bench_v2full_strong_1rank_rccl.json | tail -c 1800; tail -3 gpurun_out/r04_bench_v2full_dist1.err
