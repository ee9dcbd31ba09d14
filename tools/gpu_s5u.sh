#!/bin/bash
# round 5, trip u: bf16 attention with 16-wave workgroups (512 queries per staged K / V tile)
TAG=${1:-s5u}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for v in 2 16 2 16; do NOMAD_BF16_ATTN_V3=$v timeout 300 python3 tools/attn_bf16_ab.py >> $OUT/attn_ab.jsonl 2>> $OUT/attn_ab.err; done
cat $OUT/attn_ab.jsonl; tail -2 $OUT/attn_ab.err
for rep in 1 2; do for v in 2 16; do
  NOMAD_DIAG_LIB=1 NOMAD_BF16_ATTN_V3=$v timeout 600 python bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $OUT/bench_c5_$v_$rep.json 2> $OUT/bench_c5_$v_$rep.err
  echo "ATTN_V3=$v rep $rep: $(python3 -c "import json,sys; d=json.load(open('$OUT/bench_c5_$v_$rep.json')); print(d['value'], d['ms_per_step'])")" | tee -a $OUT/summary.txt
done; done
