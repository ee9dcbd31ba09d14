#!/bin/bash
# round 5, trip n: residual GEMMs (out_proj, fc2) on the persistent kernel vs on the one-tile-per-workgroup kernel inside the C5 forward
TAG=${1:-s5n}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2 3; do for v in 1 0; do
  NOMAD_DIAG_LIB=1 NOMAD_BF16_P9_RES=$v timeout 600 python bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $OUT/bench_c5_$v_$rep.json 2> $OUT/bench_c5_$v_$rep.err
  echo "P9_RES=$v rep $rep: $(python3 -c "import json,sys; d=json.load(open('$OUT/bench_c5_$v_$rep.json')); print(d['value'], d['ms_per_step'])")" | tee -a $OUT/summary.txt
done; done
