#!/bin/bash
# Per-launch durations IN ORDER (rocprofv3 --kernel-trace) of the packed square-root-information kernel and the packed-R Hessian
# sweep: the stats tables show 344-543 us between the fastest and slowest launch of one run -- is it periodic with the batch pool?
R=${GRAFT_REPO_ROOT:-$PWD}; export TMPDIR=/tmp
mkdir -p $R/gpurun_out; O=$R/gpurun_out/r06_seq.txt; : > $O
for spec in sqrt_info_packed:1000000:0:40 factor_v1_hessian_tri:1000000:0:40 sqrt_info:1000000:0:40; do
  D=/tmp/seq_$$_${spec%%:*}; mkdir -p $D; cd /tmp
  timeout 600 rocprofv3 --kernel-trace -d $D -o kt -- python $R/tools/microbench.py $spec > $D/out.txt 2>&1
  grep launch_us $D/out.txt >> $O
  python - "$D" >> $O <<'PY'
import glob, sqlite3, sys
for db in sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True)):
    cur = sqlite3.connect(db).cursor()
    cols = [d[1] for d in cur.execute("pragma table_info(kernels)").fetchall()]
    g = lambda *n: next(x for x in n if x in cols)
    rows = cur.execute("select %s, %s, %s from kernels order by %s" % (g("name", "kernel_name"), g("start", "start_timestamp"), g("end", "end_timestamp"), g("start", "start_timestamp"))).fetchall()
    sel = [(s, e) for n, s, e in rows if "sqrt_info_kernel" in n or "hessian_kernel" in n]
    print("launches", len(sel))
    print(" ".join("%.0f" % ((e - s) / 1e3) for s, e in sel))
    print("gaps us:", " ".join("%.0f" % ((sel[i + 1][0] - sel[i][1]) / 1e3) for i in range(min(len(sel) - 1, 60))))
PY
  rm -rf $D
done
cat $O
