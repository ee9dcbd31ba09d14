#!/bin/bash
# full GPU suite + default bench (+ rocprofv3 kernel stats of it).  Usage: bash tools/gpu_full.sh <tag>
TAG=${1:-full}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu -x --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -n 5 $OUT/pytest_gpu.log
timeout 900 python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" | tee -a $OUT/summary.txt
python3 - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print({k: d[k] for k in ("value", "ms_per_step", "kernel_time_ms_per_step")}, d["roofline"]["frac"], d.get("also_measured", {}).get("value"), d.get("also_measured_c5", {}).get("value"))
PY
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o prof -- python3 $ROOTDIR/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --single-stream --no-also > $OUT/prof_bench.json 2> $OUT/prof.err); echo "rocprof exit $?" | tee -a $OUT/summary.txt
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv && head -12 $OUT/kernel_stats.csv | cut -c1-180
find $OUT/prof -type f ! -name "*stats*" -delete 2>/dev/null
