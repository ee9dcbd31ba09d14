#!/bin/bash
# two tile shapes in one launch (gemm_f32_mixed_kernel, diag tile 99): bit check and A/B per shape
TAG=${1:-r5a}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python3 tools/dbg_tr.py 99 > $OUT/dbg_tr.txt 2>&1
grep -v "amdgpu.ids" $OUT/dbg_tr.txt | cut -c1-200
timeout 900 python3 tools/gemm_ab.py --tiles 33,97,99,-1 --shapes out,fc2,out_h,fc2_h --rounds 4 > $OUT/gemm_ab_res.jsonl 2> $OUT/gemm_ab_res.err
timeout 900 python3 tools/gemm_ab.py --tiles 33,90,99,-1 --shapes fc1,fc1_h,qkv_h --rounds 4 > $OUT/gemm_ab_dir.jsonl 2> $OUT/gemm_ab_dir.err
python3 - <<PY
import json
for f in ("$OUT/gemm_ab_res.jsonl", "$OUT/gemm_ab_dir.jsonl"):
    for l in open(f):
        d = json.loads(l)
        print(d.get("shape"), d.get("tile"), d.get("ms_med"), d.get("tflops"), d.get("tflops_best"), d.get("bit_identical"), d.get("skipped", ""))
PY
