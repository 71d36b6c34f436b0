cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out; O=gpurun_out/r06_preramp.txt; : > $O
for round in 1 2 3; do
for ms in 0 60 150 400; do
  CPI_BENCH_PRERAMP_MS=$ms python bench.py --steps 20 --warmup 5 --no-extra --no-cpu 2>/dev/null | tail -n 1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('preramp_ms', $ms, 'value %.4g' % d['value'], 'ms_per_step %.5f' % d['ms_per_step'], 'launch_us %.2f' % d['roofline']['launch_us'], 'frac %.4f' % d['roofline']['frac'])" >> $O
done
done
cat $O
