#!/bin/bash
# One GPU-box round trip: smoke, GPU tests, a short bench, a rocprofv3 kernel-stats pass.
# Usage (from the repo root on the GPU box):  bash tools/gpu_check.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/summary.txt
timeout 1500 python -m pytest tests -q -m gpu -x --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -n 40 $OUT/pytest_gpu.log
timeout 900 python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" | tee -a $OUT/summary.txt
cat $OUT/bench.json
ROOTDIR=$(pwd)
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/$OUT/prof -o prof -- python3 $ROOTDIR/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --single-stream --no-also > $ROOTDIR/$OUT/prof_bench.json 2> $ROOTDIR/$OUT/prof.err); echo "rocprof exit $?" | tee -a $OUT/summary.txt
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -r head -n 30
# keep only the small summaries (gpurun_out merge limit is 64 MiB)
find $OUT/prof -name "*kernel_trace*.csv" -size +8M -delete
