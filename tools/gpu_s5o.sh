#!/bin/bash
# round 5, trip o: vendor yardstick at the end of the round (bf16: persistent kernel; fp32: the tiles the dispatch picks, full and half batch)
TAG=${1:-s5o}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python3 tools/lib_gemm_yardstick.py --f32 "" > $OUT/yardstick_bf16.jsonl 2> $OUT/y1.err; echo "bf16 exit $?"
timeout 900 python3 tools/gemm_ab.py --tiles 33,31,-1 --shapes qkv,out,fc1,fc2,conv3,qkv_h,fc1_h,out_h,fc2_h --iters 8 --rounds 3 > $OUT/gemm_ab_f32.jsonl 2> $OUT/y2.err; echo "f32 exit $?"
python3 - <<PY
import json
for l in open("$OUT/yardstick_bf16.jsonl"):
    d = json.loads(l); print(d["shape"], round(d["lib_tflops"],1), round(d.get("ours_tflops",0),1))
for l in open("$OUT/gemm_ab_f32.jsonl"):
    d = json.loads(l); print(d["shape"], d["tile"], d["tflops"], d["tflops_best"])
PY
