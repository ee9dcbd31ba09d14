#!/bin/bash
# round 4, trip 9: persistent kernel with load / multiply cursors (82), the same kernel one tile per workgroup (87)
TAG=${1:-r4i}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python3 tools/dbg_tr.py 82,87 > $OUT/dbg_tr.txt 2>&1
timeout 900 python3 tools/gemm_ab.py --tiles 33,72,84,82,87,83,-1 --shapes qkv,out,fc1,fc2,conv3 > $OUT/gemm_ab.jsonl 2> $OUT/gemm_ab.err
echo "gemm_ab exit $?" | tee -a $OUT/summary.txt
cat $OUT/dbg_tr.txt | cut -c1-200; cat $OUT/gemm_ab.jsonl | cut -c1-150; tail -5 $OUT/gemm_ab.err
