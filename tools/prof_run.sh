#!/bin/bash
# usage: tools/prof_run.sh <out.md> [bench args...]   (run on the GPU box; keeps only the markdown table)
R=$PWD; OUT=$1; shift
export TMPDIR=/tmp
D=/tmp/prof_$$; mkdir -p $D; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $D -o kt -- python $R/bench.py --no-cpu "$@" > $D/bench.out 2> $D/bench.err
python $R/tools/kernel_stats.py "$D/**/*.db" > $R/$OUT
tail -1 $D/bench.out > $R/${OUT%.md}_bench.json
rm -rf $D
cat $R/$OUT
