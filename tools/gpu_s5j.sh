#!/bin/bash
# round 5, trip j: persistent launches of the two half-batches sharing the CUs (grid = CUs / parts) vs queueing; 2 / 3 / 4 parts
TAG=${1:-s5j}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do for cfg in "2 0" "2 1" "3 1" "4 1" "3 0"; do
  set -- $cfg
  NOMAD_DIAG_LIB=1 NOMAD_BF16_SPLIT_WAYS=$1 NOMAD_BF16_P9_SHARE=$2 timeout 600 python bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $OUT/bench_c5_$1$2_$rep.json 2> $OUT/bench_c5_$1$2_$rep.err
  echo "WAYS=$1 SHARE=$2 rep $rep: $(python3 -c "import json,sys; d=json.load(open('$OUT/bench_c5_$1$2_$rep.json')); print(d['value'], d['ms_per_step'])")" | tee -a $OUT/summary.txt
done; done
