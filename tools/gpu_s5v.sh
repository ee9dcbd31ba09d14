#!/bin/bash
# round 5, trip v: where in the K tile the persistent GEMM issues its B DMA (tile 60: phases 1 / 2; 65: phase 3; 66: phases 2 / 3)
TAG=${1:-s5v}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python3 tools/p9_ab.py --tiles 60,67,65 --vendor 0 --shapes c5_qkv,c5_out,c5_fc1,c5_fc2,c5_conv4,c5_fc1_nogelu,sq4096 --rounds 4 > $OUT/p9_ab.jsonl 2> $OUT/p9_ab.err; echo "exit $?"
python3 - <<PY
import json
for l in open("$OUT/p9_ab.jsonl"):
    d = json.loads(l)
    print(d["shape"], {t: (d[f"tile{t}_us"], d[f"tile{t}_tf"], d[f"tile{t}_bit_identical_to_60"]) for t in (60, 67, 65)})
PY
tail -3 $OUT/p9_ab.err
