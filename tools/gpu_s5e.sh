#!/bin/bash
# round 5, trip e: C5 bench - persistent GEMM x two-stream split x tail row-split
TAG=${1:-s5e}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python3 tools/p9_ab.py --tiles 58,64,60 --shapes c5_out,c5_fc2,c5h_out,c5h_fc2 > $OUT/p9_ab.jsonl 2> $OUT/p9_ab.err; echo "p9_ab exit $?" | tee -a $OUT/summary.txt
cat $OUT/p9_ab.jsonl; tail -3 $OUT/p9_ab.err
for rep in 1 2; do for cfg in "4000 0 0" "4000 1 0" "4000 1 1" "0 1 0" "0 1 1" "0 0 0"; do
  set -- $cfg
  NOMAD_DIAG_LIB=1 NOMAD_BF16_SPLIT_ROWS=$1 NOMAD_BF16_P9=$2 NOMAD_BF16_P9_TAIL=$3 timeout 600 python bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $OUT/bench_c5_$1_$2$3_$rep.json 2> $OUT/bench_c5_$1_$2$3_$rep.err
  echo "SPLIT=$1 P9=$2 TAIL=$3 rep $rep: $(python3 -c "import json,sys; d=json.load(open('$OUT/bench_c5_$1_$2$3_$rep.json')); print(d['value'], d['ms_per_step'])")" | tee -a $OUT/summary.txt
done; done
