#!/bin/bash
# round 4, trip 6: set-up detail of the production tile; skewed 8-wave + direct epilogue (84); direct epilogue after the residual re-pipelining (72)
TAG=${1:-r4f}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python3 tools/dbg_tr.py 72,84 > $OUT/dbg_tr.txt 2>&1
timeout 600 python3 tools/gemm_timeline_f32.py --shapes qkv --tiles 86 > $OUT/timeline.jsonl 2> $OUT/timeline.err
timeout 900 python3 tools/gemm_ab.py --tiles 33,72,84,-1 --shapes qkv,out,fc1,fc2,conv3 > $OUT/gemm_ab.jsonl 2> $OUT/gemm_ab.err
echo "gemm_ab exit $?" | tee -a $OUT/summary.txt
cat $OUT/dbg_tr.txt | cut -c1-200; cat $OUT/timeline.jsonl; cat $OUT/gemm_ab.jsonl; tail -5 $OUT/gemm_ab.err
