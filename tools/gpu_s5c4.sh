#!/bin/bash
# round 5: kernel trace of configs[3] split into forward / backward (tools/c4_trace_table.py)
TAG=${1:-s5c4}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o c4 -- python3 $ROOTDIR/tools/c4_profile.py 20 > $OUT/c4_profile.txt 2> $OUT/prof.err); echo "rocprof exit $?" | tee -a $OUT/summary.txt
cat $OUT/c4_profile.txt
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python3 tools/c4_trace_table.py $f > $OUT/c4_trace_table.txt 2>&1; cat $OUT/c4_trace_table.txt
rm -rf $OUT/prof
