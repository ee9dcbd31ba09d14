#!/bin/bash
# residual-ahead epilogue (OPT bit 32768) + bias-first direct epilogue: bit check, A/B per shape, A/B on the bench
TAG=${1:-r4w}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python3 tools/dbg_tr.py 72,84,89,90,91,92,93,97,98 > $OUT/dbg_tr.txt 2>&1
grep -c "ndiff 0 " $OUT/dbg_tr.txt; grep -v "ndiff 0 " $OUT/dbg_tr.txt | head -20
timeout 900 python3 tools/gemm_ab.py --tiles 33,91,97,31,92,98,-1 --shapes out,fc2 --rounds 4 > $OUT/gemm_ab_res.jsonl 2> $OUT/gemm_ab_res.err
timeout 900 python3 tools/gemm_ab.py --tiles 33,84,90,-1 --shapes qkv,fc1,conv3 --rounds 4 > $OUT/gemm_ab_dir.jsonl 2> $OUT/gemm_ab_dir.err
python3 - <<PY
import json
for f in ("$OUT/gemm_ab_res.jsonl", "$OUT/gemm_ab_dir.jsonl"):
    for l in open(f):
        d = json.loads(l)
        print(d.get("shape"), d.get("tile"), d.get("ms_med"), d.get("tflops"), d.get("tflops_best"), d.get("bit_identical"), d.get("skipped", ""))
PY
for rep in 1 2; do for ra in 1 0; do
  NOMAD_F32_RES_AHEAD=$ra timeout 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-profile --live-traffic off > $OUT/b.json 2> $OUT/bench.err
  python3 -c "
import json
d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1])
print('res_ahead $ra', d['value'], d['ms_per_step'], d['roofline']['frac'])
"
done; done | tee $OUT/ab_bench.txt
