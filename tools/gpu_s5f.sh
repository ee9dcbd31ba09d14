#!/bin/bash
# round 5, trip f: kernel trace of the C5 forward (single stream), persistent GEMM + 16-wide attention on / off -> per-shape table;
# graphed-loss test; GPU suite subset
TAG=${1:-s5f}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_race_screen.py -q -m gpu -x --timeout 900 > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -n 15 $OUT/pytest.log
export NOMAD_DIAG_LIB=1
for cfg in "1 1 0" "0 0 0" "1 1 1"; do
  set -- $cfg
  export NOMAD_BF16_P9=$1 NOMAD_BF16_ATTN_V3=$2 NOMAD_BF16_P9_TAIL=$3
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$1$2$3 -o c5 -- python3 $ROOTDIR/bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 5 --warmup 2 --no-cpu-baseline --no-profile --single-stream > $OUT/prof_bench_$1$2$3.json 2> $OUT/prof_$1$2$3.err); echo "rocprof $cfg exit $?" | tee -a $OUT/summary.txt
  f=$(find $OUT/prof_$1$2$3 -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python3 tools/c5_layer_table.py $f > $OUT/layer_table_$1$2$3.json && cat $OUT/layer_table_$1$2$3.json
  g=$(find $OUT/prof_$1$2$3 -name "*kernel_stats.csv" | head -1); [ -n "$g" ] && cp $g $OUT/kernel_stats_$1$2$3.csv
  rm -rf $OUT/prof_$1$2$3
done
