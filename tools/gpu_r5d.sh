#!/bin/bash
# per-launch durations of the fp32 GEMMs inside the headline forward (rocprofv3 --kernel-trace), grouped by instantiation and
# grid, for the round-4 variants on / off.  Usage: bash tools/gpu_f32_prof.sh <tag> [extra bench args, e.g. --single-stream]
TAG=${1:-f32prof}; shift
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for m in 1 0; do
  export NOMAD_F32_MIXED=$m
  rm -rf $OUT/prof_$m
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_$m -o f32 -- python3 $ROOTDIR/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile --no-also --live-traffic off "$@" > $OUT/prof_$m.json 2> $OUT/prof_$m.err; echo "rocprof variants=$m exit $?"
  t=$(find $OUT/prof_$m -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python3 - "$t" > $OUT/per_launch_$m.txt <<'PY'
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "gemm_f32" not in n and "attention" not in n and "layernorm" not in n and "conv0" not in n: continue
    m = re.search(r"(gemm_f32_\w+)<([^>]*)>", n)
    key = (m.group(1)[9:] + "<" + m.group(2).replace(" ", "") + ">") if m else n.split("(")[0][:40]
    g = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)
    per[(key, g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    v.sort(); tot += sum(v)
    print(f"{k[0]:<60} grid {k[1]:>9}  launches {len(v):>4}  median {v[len(v)//2]:>9.1f} us  sum {sum(v)/1e3:>9.3f} ms")
print("total %.3f ms" % (tot / 1e3))
PY
  find $OUT/prof_$m -type f -delete 2>/dev/null
  echo "== variants=$m"; head -30 $OUT/per_launch_$m.txt
done
