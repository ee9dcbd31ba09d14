#!/bin/bash
# which variant of the 256x128 fp32 kernel loses on the "slow" boxes?  production dispatch (33) / lean only (88) / lean + direct (89) /
# lean + skew + direct (90) / lean + skew, LDS epilogue (91) / skew + direct, general set-up (84) / plain P instantiation (env) / vendor
TAG=${1:-r5q}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
show() { python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if 'tflops' in d: print('$1', d['shape'], d['tile'], d['tflops'], d['tflops_best'])
"; }
timeout 400 python3 tools/gemm_ab.py --tiles 33,88,89,90,91,84,-1 --shapes qkv,conv3 --rounds 3 2>/dev/null | show variants | tee $OUT/variants.txt
NOMAD_F32_LEAN=0 timeout 300 python3 tools/gemm_ab.py --tiles 33,-1 --shapes qkv,conv3 --rounds 3 2>/dev/null | show lean_off | tee -a $OUT/variants.txt
timeout 120 python3 tools/clock_under_load.py 2>/dev/null | tail -1 | tee -a $OUT/variants.txt
rocm-smi --showclocks --showpower --showmaxpower 2>/dev/null | grep -i "clk\|power" | tee -a $OUT/variants.txt
