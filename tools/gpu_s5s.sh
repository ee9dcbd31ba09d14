#!/bin/bash
# round 5, trip s: bf16 attention with the scalar-base tile fetch: tests, alone, C5 bench A/B against v2
TAG=${1:-s5s}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_precision_vs_oracle.py tests/test_gpu_race_screen.py -q -m gpu -x --timeout 900 > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -n 3 $OUT/pytest.log
for v in 0 2 0 2; do NOMAD_BF16_ATTN_V3=$v timeout 300 python3 tools/attn_bf16_ab.py >> $OUT/attn_ab.jsonl 2>> $OUT/attn_ab.err; done
cat $OUT/attn_ab.jsonl
for rep in 1 2 3; do for cfg in "0 0" "1 0" "1 2"; do
  set -- $cfg
  NOMAD_DIAG_LIB=1 NOMAD_BF16_P9=$1 NOMAD_BF16_ATTN_V3=$2 timeout 600 python bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $OUT/bench_c5_$1$2_$rep.json 2> $OUT/bench_c5_$1$2_$rep.err
  echo "P9=$1 ATTN_V3=$2 rep $rep: $(python3 -c "import json,sys; d=json.load(open('$OUT/bench_c5_$1$2_$rep.json')); print(d['value'], d['ms_per_step'])")" | tee -a $OUT/summary.txt
done; done
