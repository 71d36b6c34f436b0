#!/bin/bash
# Launch-by-launch durations of the headline kernel (10 000 windows x 50, graph replay of 2000 steps): is the 11 - 18 us spread of the
# stats table periodic with the 12-batch pool, or a clock transient?
R=${GRAFT_REPO_ROOT:-$PWD}; export TMPDIR=/tmp
mkdir -p $R/gpurun_out; O=$R/gpurun_out/r06_seq_headline.txt; : > $O
D=/tmp/seqh_$$; mkdir -p $D; cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $D -o kt -- python $R/bench.py --steps 2000 --warmup 200 --no-extra --no-cpu > $D/out.txt 2>&1
tail -n 1 $D/out.txt | cut -c1-200 >> $O
python - "$D" >> $O <<'PY'
import glob, sqlite3, sys
for db in sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True)):
    cur = sqlite3.connect(db).cursor()
    cols = [d[1] for d in cur.execute("pragma table_info(kernels)").fetchall()]
    g = lambda *n: next(x for x in n if x in cols)
    rows = cur.execute("select %s, %s, %s from kernels order by %s" % (g("name", "kernel_name"), g("start", "start_timestamp"), g("end", "end_timestamp"), g("start", "start_timestamp"))).fetchall()
    sel = [(s, e) for n, s, e in rows if "cpi_mean_kernel" in n]
    print("launches", len(sel))
    d = [(e - s) / 1e3 for s, e in sel]
    tail = d[-2000:]
    print("last 2000: mean %.2f min %.2f max %.2f" % (sum(tail) / len(tail), min(tail), max(tail)))
    # by position in the 12-batch pool
    for k in range(12):
        v = tail[k::12]
        print("pool slot %2d: mean %.2f min %.2f max %.2f" % (k, sum(v) / len(v), min(v), max(v)))
    print("first 120 of the last 2000:", " ".join("%.1f" % x for x in tail[:120]))
    gaps = [(sel[i + 1][0] - sel[i][1]) / 1e3 for i in range(len(sel) - 2000, len(sel) - 1)]
    print("gaps: mean %.2f min %.2f max %.2f" % (sum(gaps) / len(gaps), min(gaps), max(gaps)))
PY
rm -rf $D
cat $O
