#!/bin/bash
# round 4, trip 10: lean set-up in the main template (88-93) against production / persistent / vendor
TAG=${1:-r4j}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python3 tools/dbg_tr.py 88,89,90,91,92,93 > $OUT/dbg_tr.txt 2>&1
timeout 900 python3 tools/gemm_ab.py --tiles 33,88,89,90,91,87,31,92,93,-1 --shapes qkv,out,fc1,fc2,conv3,conv5 --rounds 4 > $OUT/gemm_ab.jsonl 2> $OUT/gemm_ab.err
echo "gemm_ab exit $?" | tee -a $OUT/summary.txt
grep -c "ndiff 0 " $OUT/dbg_tr.txt; grep -v "ndiff 0 " $OUT/dbg_tr.txt | cut -c1-200; tail -5 $OUT/gemm_ab.err
python3 - <<PY
import json
rows=[json.loads(l) for l in open("$OUT/gemm_ab.jsonl")]
shapes=[]; tiles=[]
for r in rows:
    if r['shape'] not in shapes: shapes.append(r['shape'])
    if r['tile'] not in tiles: tiles.append(r['tile'])
print("tile   "+" ".join(f"{s:>9}" for s in shapes))
for t in tiles:
    line=f"{t:>4}  "
    for s in shapes:
        m=[r for r in rows if r['shape']==s and r['tile']==t]
        line+= f" {m[0]['tflops']:>6.1f}{'*' if m[0]['bit_identical'] else ' '} " if m else "     -    "
    print(line)
PY
