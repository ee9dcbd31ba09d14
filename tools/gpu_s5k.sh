#!/bin/bash
# round 5, trip k: fp32 GEMM tile choice for the half-batch N = 768 shapes (what Engine.embed launches): 128x128x32 (31, shipped),
# 128x64x32 3-stage (34), 128x64x32 2-stage (29), 256x128 / two-shape (33), vendor (-1)
TAG=${1:-s5k}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python3 tools/gemm_ab.py --tiles 31,34,29,33,-1 --shapes fc2_h,out_h,proj_h,conv6_h --iters 8 --rounds 3 > $OUT/gemm_ab.jsonl 2> $OUT/gemm_ab.err; echo "exit $?"
cat $OUT/gemm_ab.jsonl | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print(d)
"
tail -3 $OUT/gemm_ab.err
