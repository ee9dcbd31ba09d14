#!/bin/bash
# round 4, trip 5: the persistent kernel (tile 82 / 83) and the fixed direct epilogue (72) against production and the vendor
TAG=${1:-r4e}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python3 tools/dbg_tr.py 72,73,77,82 > $OUT/dbg_tr.txt 2>&1
timeout 900 python3 tools/gemm_ab.py --tiles 33,72,82,83,-1 --shapes qkv,out,fc1,fc2,conv3 > $OUT/gemm_ab.jsonl 2> $OUT/gemm_ab.err
echo "gemm_ab exit $?" | tee -a $OUT/summary.txt
cat $OUT/dbg_tr.txt | cut -c1-300; cat $OUT/gemm_ab.jsonl; tail -5 $OUT/gemm_ab.err
