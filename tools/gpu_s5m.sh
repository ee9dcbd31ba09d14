#!/bin/bash
# round 5, trip m: full GPU suite, smoke, default bench, rocprofv3 kernel stats of the headline (single-stream timing pass) and of configs[4]
TAG=${1:-s5m}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -q -m gpu -x --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -n 4 $OUT/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/summary.txt; tail -2 $OUT/smoke.log
timeout 900 python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" | tee -a $OUT/summary.txt
python3 - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print({k: d.get(k) for k in ("value", "ms_per_step")}, d["roofline"]["frac"], "x3", d.get("also_measured", {}).get("value"), "c5", d.get("also_measured_c5", {}).get("value"))
print("c4", {k: v for k, v in d.get("also_measured_c4", {}).items() if "ms" in k or "graph" in k})
print("attn ms", d.get("kernel_time_ms_per_step", {}).get("attention_mfma"))
PY
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o prof -- python3 $ROOTDIR/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --single-stream --no-also > $OUT/prof_bench.json 2> $OUT/prof.err); echo "rocprof exit $?" | tee -a $OUT/summary.txt
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv && head -8 $OUT/kernel_stats.csv | cut -c1-180
rm -rf $OUT/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5 -o c5 -- python3 $ROOTDIR/bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 5 --warmup 2 --no-cpu-baseline --no-profile > $OUT/prof_bench_c5.json 2> $OUT/prof_c5.err); echo "rocprof c5 exit $?" | tee -a $OUT/summary.txt
f=$(find $OUT/prof_c5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/c5_kernel_stats.csv && head -6 $OUT/c5_kernel_stats.csv | cut -c1-160
rm -rf $OUT/prof_c5
