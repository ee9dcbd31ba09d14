#!/bin/bash
# configs[4] (bf16, 30 s clips) kernel table: rocprofv3 --kernel-trace --stats of the bench line, single stream and default
TAG=${1:-r5n}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for mode in single default; do
  extra=""; [ $mode = single ] && extra="--single-stream"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$mode -o c5 -- python3 $ROOTDIR/bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 4 --warmup 2 --no-cpu-baseline --no-also --no-profile --live-traffic off $extra > $OUT/b_$mode.json 2> $OUT/err_$mode.txt)
  f=$(find $OUT/prof_$mode -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/c5_kernel_stats_$mode.csv
  find $OUT/prof_$mode -type f ! -name "*stats*" -delete 2>/dev/null
  echo "== $mode"; python3 - <<PY
import csv, json
d = json.loads(open("$OUT/b_$mode.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])
rows = list(csv.DictReader(open("$OUT/c5_kernel_stats_$mode.csv")))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("kernel time per step %.2f ms (6 steps)" % (tot / 1e6 / 6))
for r in rows[:12]:
    print("%-100s calls %5s  %7.2f ms/step  avg %8.1f us" % (r["Name"][:100], r["Calls"], int(r["TotalDurationNs"]) / 1e6 / 6, float(r["AverageNs"]) / 1e3))
PY
done
