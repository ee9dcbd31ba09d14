#!/bin/bash
TAG=${1:-r4o}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python3 tools/dbg_tr.py 90,96 > $OUT/dbg_tr.txt 2>&1
timeout 900 python3 tools/gemm_ab.py --tiles 33,90,82,96,87,-1 --shapes qkv,fc1,conv3,fc2 --rounds 4 > $OUT/gemm_ab.jsonl 2> $OUT/gemm_ab.err
grep -c "ndiff 0 " $OUT/dbg_tr.txt; grep -v "ndiff 0 " $OUT/dbg_tr.txt | cut -c1-200 | head -5
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
