#!/bin/bash
# 16x16x4 MFMA products in every fp32 GEMM instantiation: kernel tests, rates, clock, bench
TAG=${1:-r6a}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_gpu_kernels.py -q -x -k "gemm" 2>&1 | tail -4
timeout 500 python3 tools/gemm_ab.py --tiles 33,31,37,34,20,-1 --shapes qkv,conv3,fc1,fc2,out --rounds 3 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if 'tflops' in d: print(d['shape'], d['tile'], d['tflops'], d['tflops_best'], d['bit_identical'])
" | tee $OUT/ab.txt
timeout 200 python3 tools/clock_under_load.py 2>/dev/null | tail -1 | tee -a $OUT/ab.txt
for i in 1 2; do timeout 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-profile --live-traffic off 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])
"; done | tee -a $OUT/ab.txt
