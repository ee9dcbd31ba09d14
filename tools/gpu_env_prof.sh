#!/bin/bash
# per-launch durations of the bf16 GEMMs inside the C5 forward (single stream, rocprofv3 --kernel-trace) for two settings of one
# environment switch.  Usage: bash tools/gpu_env_prof.sh <tag> <ENV_NAME> <value A> <value B>
TAG=${1:-envprof}; VAR=$2; A=$3; B=$4
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for m in $A $B $A $B $5; do
  export $VAR=$m
  rm -rf $OUT/prof_$m
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_$m -o c5 -- python3 $ROOTDIR/bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 5 --warmup 2 --no-cpu-baseline --no-profile --single-stream > $OUT/prof_$m.json 2> $OUT/prof_$m.err; echo "rocprof $VAR=$m exit $?"
  t=$(find $OUT/prof_$m -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python3 - "$t" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
seq = [(r["Kernel_Name"], int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0) // 512, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if "8phase" in r["Kernel_Name"]]
per = collections.defaultdict(list)
for n, g, d in seq[-55 * 3:]:
    key = (g, "short" if d < 150 else "long") if g == 564 else (g, "")
    per[key].append(d)
tot = 0
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    v.sort(); tot += sum(v)
    print(k, "launches", len(v), "median us", round(v[len(v) // 2], 1), "sum ms", round(sum(v) / 1e3, 3))
print("all 8-phase launches of the last 3 steps: %.3f ms per step" % (tot / 3e3))
PY
  find $OUT/prof_$m -type f -delete 2>/dev/null
done
