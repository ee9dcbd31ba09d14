#!/bin/bash
# round 4, trip 3: direct epilogue from transposed accumulators (tiles 72-77) A/B, its timeline, prologue detail of production
TAG=${1:-r4c}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python3 tools/gemm_ab.py --tiles 33,72,77,74,31,73,-1 --shapes qkv,out,fc1,fc2,conv3 > $OUT/gemm_ab.jsonl 2> $OUT/gemm_ab.err
echo "gemm_ab exit $?" | tee -a $OUT/summary.txt
timeout 600 python3 tools/gemm_timeline_f32.py --shapes qkv,fc1 --tiles 68,75,76 > $OUT/timeline.jsonl 2> $OUT/timeline.err
echo "timeline exit $?" | tee -a $OUT/summary.txt
cat $OUT/gemm_ab.jsonl; cat $OUT/timeline.jsonl; tail -3 $OUT/timeline.err $OUT/gemm_ab.err
