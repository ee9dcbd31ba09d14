#!/bin/bash
# round 4, trip 4: where the direct epilogue's bits differ; raised wave priority for set-up / epilogue (tiles 78-81)
TAG=${1:-r4d}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python3 tools/dbg_tr.py > $OUT/dbg_tr.txt 2>&1
timeout 900 python3 tools/gemm_ab.py --tiles 33,78,72,79,-1 --shapes qkv,out,fc1,fc2,conv3 > $OUT/gemm_ab.jsonl 2> $OUT/gemm_ab.err
echo "gemm_ab exit $?" | tee -a $OUT/summary.txt
timeout 600 python3 tools/gemm_timeline_f32.py --shapes qkv,fc1 --tiles 80,81 > $OUT/timeline.jsonl 2> $OUT/timeline.err
echo "timeline exit $?" | tee -a $OUT/summary.txt
cat $OUT/dbg_tr.txt | cut -c1-700; cat $OUT/gemm_ab.jsonl; cat $OUT/timeline.jsonl
