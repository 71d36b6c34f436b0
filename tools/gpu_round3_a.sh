#!/bin/bash
# Round-3 session A: bench.py after the rewrite -- default command, eager vs graph, distributed paths on one GPU.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_group.py tests/test_gpu_bench.py -x -q 2>&1 | tail -15 > gpurun_out/r03_pytest_b.txt; cat gpurun_out/r03_pytest_b.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; tail -c 1500 gpurun_out/r03_bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_bench_default.json').read().strip().splitlines()[-1])
print("HEADLINE", d["value"], d["ms_per_step"], d["config"]["launch_mode"], d["roofline"]["launch_us"], d["roofline"]["frac"])
for r in d.get("extra", []):
    if "error" in r: print("ERR", r); continue
    print("%-22s W=%-8s %10.4g/s launch_ms=%8.4f mode=%s frac=%.3f cpu=%s asm=%s" % (r["workload"], r.get("units_per_step"), r["value"], r.get("launch_ms", 0), r.get("launch_mode"),
          r.get("roofline", {}).get("frac", 0), (r.get("cpu_baseline") or {}).get("value"), (r.get("assembly") or {}).get("ms_per_batch")))
PY
for mode in "" "--eager"; do for k in "20 5" "2000 200"; do set -- $k; python bench.py --steps $1 --warmup $2 --no-extra --no-cpu $mode | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mode=%-8s K=%-5d us/step wall=%.3f kernel=%.3f' % ('$mode' or 'graph', d['steps'], d['ms_per_step']*1e3, d['roofline']['launch_us']))"; done; done
CPI_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu > gpurun_out/r03_bench_dist1.json 2> gpurun_out/r03_bench_dist1.err; tail -c 1200 gpurun_out/r03_bench_dist1.json; tail -3 gpurun_out/r03_bench_dist1.err
CPI_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29532 bench.py --gpus 1 --steps 20 --warmup 5 --workload v2_full --scaling strong --no-extra --no-cpu > gpurun_out/r03_bench_v2full_dist1.json 2> gpurun_out/r03_bench_v2full_dist1.err; tail -c 1200 gpurun_out/r03_bench_v2full_dist1.json; tail -3 gpurun_out/r03_bench_v2full_dist1.err
