#!/bin/bash
# round 5, trip h: full GPU suite, default bench, C5 kernel stats (two-stream default) + PMC passes of the persistent GEMM, clock under C5
TAG=${1:-s5h}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -q -m gpu -x --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -n 8 $OUT/pytest_gpu.log
timeout 900 python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" | tee -a $OUT/summary.txt
python3 - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "kernel_time_ms_per_step")}, d["roofline"]["frac"], d.get("also_measured", {}).get("value"), d.get("also_measured_c5", {}).get("value"))
print("c4", {k: v for k, v in d.get("also_measured_c4", {}).items() if "ms" in k or "graph" in k})
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("torch_default_threads"))
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5 -o c5 -- python3 $ROOTDIR/bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 5 --warmup 2 --no-cpu-baseline --no-profile > $OUT/prof_bench_c5.json 2> $OUT/prof_c5.err); echo "rocprof c5 exit $?" | tee -a $OUT/summary.txt
f=$(find $OUT/prof_c5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/c5_kernel_stats.csv && head -8 $OUT/c5_kernel_stats.csv | cut -c1-160
rm -rf $OUT/prof_c5
for v in 4000 0; do NOMAD_BF16_SPLIT_ROWS=$v timeout 300 python3 tools/clock_c5.py $([ $v = 0 ] && echo single || echo split) >> $OUT/clock_c5.jsonl 2>> $OUT/clock_c5.err; done
cat $OUT/clock_c5.jsonl
PMC_SET=cache bash tools/gpu_pmc_bf16.sh $TAG/pmc_qkv_p9 c5_qkv 60 > /dev/null 2>&1; cat $OUT/pmc_qkv_p9/pmc_summary.txt
PMC_SET=cache bash tools/gpu_pmc_bf16.sh $TAG/pmc_qkv_58 c5_qkv 58 > /dev/null 2>&1; cat $OUT/pmc_qkv_58/pmc_summary.txt
