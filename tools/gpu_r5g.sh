#!/bin/bash
# sweep of the 256-row part's size (NOMAD_F32_MIXED_M1) of the two-shape launch, per shape, each launch alone on the GPU
TAG=${1:-r5g}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for spec in "fc2:10752,21760,32768,43520,47104" "out:10752,21760,32768,43520,47104" "fc2_h:5376,10752,16384,21760,23552" "out_h:5376,10752,16384,21760,23552" "qkv_h:7168,14336,21760,23552" "fc1_h:10752,16384,21760,24064"; do
  shape=${spec%%:*}; list=${spec##*:}
  for m1 in ${list//,/ }; do
    NOMAD_F32_MIXED_M1=$m1 timeout 300 python3 tools/gemm_ab.py --tiles 33,99 --shapes $shape --rounds 3 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if 'tflops' in d: print('$shape', 'M1=$m1', 'tile', d['tile'], d['tflops'], d['bit_identical'])
"
  done
done | tee $OUT/sweep.txt
