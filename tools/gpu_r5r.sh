#!/bin/bash
# BK = 16 (64-byte DMA row segments) against BK = 32 (128-byte) tiles and the vendor, to compare between fast and slow boxes
TAG=${1:-r5r}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 500 python3 tools/gemm_ab.py --tiles 33,90,31,92,-1 --shapes qkv,conv3,fc1 --rounds 3 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if 'tflops' in d: print(d['shape'], d['tile'], d['tflops'], d['tflops_best'])
" | tee $OUT/bk.txt
timeout 120 python3 tools/clock_under_load.py 2>/dev/null | tail -1 | tee -a $OUT/bk.txt
