#!/bin/bash
TAG=${1:-r5h}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python3 tools/c4_profile.py 20 2>/dev/null | tee $OUT/wall.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o c4 -- python3 $ROOTDIR/tools/c4_profile.py 20 > $OUT/prof.log 2>&1)
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/c4_kernel_stats.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/c4_kernel_stats.csv")))
tot = sum(int(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print("kernel time total %.2f ms over %d launches (25 steps incl. warm-up): %.3f ms and %d launches per step" % (tot / 1e6, calls, tot / 1e6 / 25, calls // 25))
for r in rows[:14]:
    print("%-90s calls %5s total %8.2f ms avg %8.1f us" % (r["Name"][:90], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
tail -2 $OUT/prof.log
find $OUT/prof -type f ! -name "*stats*" -delete 2>/dev/null
