#!/bin/bash
# in-pipeline durations of the 256x192 / 256x256 instantiations (C5, single stream) + vendor kernel names on the hot shapes
TAG=${1:-n192prof}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for m in 0 auto; do
  if [ $m = auto ]; then unset NOMAD_BF16_N192; else export NOMAD_BF16_N192=$m; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$m -o c5 -- python3 $ROOTDIR/bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 5 --warmup 2 --no-cpu-baseline --no-profile --single-stream > $OUT/prof_$m.json 2> $OUT/prof_$m.err; echo "rocprof $m exit $?"
  f=$(find $OUT/prof_$m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_$m.csv && head -8 $OUT/kernel_stats_$m.csv | cut -c1-200
  t=$(find $OUT/prof_$m -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python3 - "$t" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# per-launch sequence of the 8-phase kernels in the LAST step: name suffix, grid size, duration
seq = [(r["Kernel_Name"], int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0) // 512, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if "8phase" in r["Kernel_Name"]]
per = collections.defaultdict(list)
for n, g, d in seq[-55 * 3:]:
    per[(n.split("<")[1].split(">")[0], g)].append(d)
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(k, "launches", len(v), "median us", round(v[len(v) // 2], 1), "sum ms", round(sum(v) / 1e3, 3))
PY
  find $OUT/prof_$m -type f ! -name "*stats*" -delete 2>/dev/null
done
unset NOMAD_BF16_N192
[ "$2" = "noyard" ] && exit 0
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_y -o y -- python3 $ROOTDIR/tools/lib_gemm_yardstick.py --iters 4 > $OUT/yardstick.jsonl 2> $OUT/err.log; echo "yard exit $?"
f=$(find $OUT/prof_y -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_yard.csv && cut -c1-300 $OUT/kernel_stats_yard.csv | head -30
find $OUT/prof_y -type f ! -name "*stats*" -delete 2>/dev/null
