#!/bin/bash
# Usage: bash tools/gpu_train_prof.sh <tag> [bench_train args]
TAG=${1:-tp}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
ROOTDIR=$(pwd)
python tools/bench_train.py "$@" --steps 4 --out $OUT/bench_train.json > $OUT/bench.log 2>&1; tail -2 $OUT/bench.log | cut -c1-1500
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/$OUT/prof -o tr -- python3 $ROOTDIR/tools/bench_train.py "$@" --steps 3 --warmup 1 > $ROOTDIR/$OUT/prof.log 2>&1
cd $ROOTDIR
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/kernel_stats.csv && head -40 $OUT/kernel_stats.csv | cut -c1-220
find $OUT/prof -type f ! -name "*stats*" -delete 2>/dev/null
