#!/bin/bash
# round 5, trip y: fp32 attention, LDS fragments as ext-vector loads (shipped) against float4 struct copies (NOMAD_F32_ATTN_STRUCT_LOADS=1:
# hipcc waits for the next tile's LDS-DMA in front of them) - the kernel alone and the headline bench, alternating
TAG=${1:-s5y}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for i in 1 2 3; do for v in 1 0; do
  echo -n "struct_loads=$v " >> $OUT/attn_f32.jsonl
  NOMAD_F32_ATTN_STRUCT_LOADS=$v timeout 300 python3 tools/attn_f32_time.py >> $OUT/attn_f32.jsonl 2>> $OUT/attn_f32.err
done; done
for v in 1 0; do
  echo -n "struct_loads=$v " >> $OUT/attn_f32.jsonl
  NOMAD_F32_ATTN_STRUCT_LOADS=$v timeout 300 python3 tools/attn_f32_time.py --B 32 --T 1499 >> $OUT/attn_f32.jsonl 2>> $OUT/attn_f32.err
done
cat $OUT/attn_f32.jsonl
for rep in 1 2 3; do for v in 1 0; do
  NOMAD_DIAG_LIB=1 NOMAD_F32_ATTN_STRUCT_LOADS=$v timeout 600 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-profile > $OUT/bench_${v}_$rep.json 2> $OUT/bench_${v}_$rep.err
  echo "struct_loads=$v rep $rep: $(python3 -c "import json,sys; d=json.load(open('$OUT/bench_${v}_$rep.json')); print(d['value'], d['ms_per_step'])")" | tee -a $OUT/summary.txt
done; done
