#!/bin/bash
# config C5 (bf16, 30 s clips, batch 32): parity tests of the bf16 path, the bench line, rocprofv3 kernel stats
# Usage: bash tools/gpu_c5.sh <tag> [notests]
TAG=${1:-c5}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
if [ "$2" != "notests" ]; then
  timeout 1200 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_precision_vs_oracle.py -q -m gpu -s --timeout 900 > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
  grep -E "vs oracle|vs fp32|passed|failed|Error|error" $OUT/pytest.log | tail -n 20
fi
timeout 600 python bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err; echo "bench exit $?" | tee -a $OUT/summary.txt
cat $OUT/bench_c5.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o c5 -- python3 $ROOTDIR/bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 5 --warmup 2 --no-cpu-baseline --no-profile --single-stream > $OUT/prof_bench.json 2> $OUT/prof.err); echo "rocprof exit $?" | tee -a $OUT/summary.txt
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv && head -25 $OUT/kernel_stats.csv | cut -c1-200
find $OUT/prof -type f ! -name "*stats*" -delete 2>/dev/null
