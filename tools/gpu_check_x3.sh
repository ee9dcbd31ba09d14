#!/bin/bash
# bf16x3 round trip: its bench line and a rocprofv3 kernel-stats pass of the same command.
TAG=${1:-x3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python bench.py --dtype bf16x3 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_x3.json 2> $OUT/bench_x3.err; echo "bench exit $?" | tee -a $OUT/summary.txt
cat $OUT/bench_x3.json
ROOTDIR=$(pwd)
# kernel stats with the two-stream batch split off: overlapping launches would inflate every kernel's duration
export NOMAD_X3_SPLIT_ROWS=0
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/$OUT/prof_x3 -o prof -- python3 $ROOTDIR/bench.py --dtype bf16x3 --steps 3 --warmup 1 --no-cpu-baseline --no-profile --single-stream > $ROOTDIR/$OUT/prof_bench_x3.json 2> $ROOTDIR/$OUT/prof_x3.err); echo "rocprof exit $?" | tee -a $OUT/summary.txt
find $OUT/prof_x3 -name "*kernel_stats*.csv" | head -1 | xargs -r head -n 16 | cut -c1-200
find $OUT/prof_x3 -name "*kernel_trace*.csv" -size +8M -delete
