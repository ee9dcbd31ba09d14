#!/bin/bash
TAG=${1:-c4prof}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o prof -- python3 $ROOTDIR/tools/bench_c4.py --steps 10 --warmup 3 > $OUT/bench_c4.json 2> $OUT/err.log
cd $ROOTDIR
cat $OUT/bench_c4.json | tail -1
head -n 25 $OUT/prof/prof_kernel_stats.csv | cut -c1-200
find $OUT/prof -name "*kernel_trace*.csv" -size +8M -delete
