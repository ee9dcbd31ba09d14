#!/bin/bash
# round 4, trip 2: per-workgroup timeline of the production fp32 GEMM (with / without epilogue stores), paced-store A/B
TAG=${1:-r4b}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python3 tools/gemm_timeline_f32.py --shapes qkv,fc1,fc2,conv3 --tiles 68,69 --dump $OUT/raw > $OUT/timeline.jsonl 2> $OUT/timeline.err
echo "timeline exit $?" | tee -a $OUT/summary.txt
timeout 900 python3 tools/gemm_ab.py --tiles 33,70,71,61,-1 --shapes qkv,fc1,out,conv3 > $OUT/gemm_ab.jsonl 2> $OUT/gemm_ab.err
echo "gemm_ab exit $?" | tee -a $OUT/summary.txt
cat $OUT/timeline.jsonl; cat $OUT/gemm_ab.jsonl; tail -3 $OUT/timeline.err
