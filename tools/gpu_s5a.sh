#!/bin/bash
# round 5, trip a: the persistent bf16 GEMM - per-shape A/B (bit-identity + time) against the shipped kernel and the vendor library,
# the bf16 test files with it on, and the C5 bench alternating NOMAD_BF16_P9=0 / 1
TAG=${1:-s5a}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python3 tools/p9_ab.py > $OUT/p9_ab.jsonl 2> $OUT/p9_ab.err; echo "p9_ab exit $?" | tee -a $OUT/summary.txt
cat $OUT/p9_ab.jsonl; tail -3 $OUT/p9_ab.err
timeout 1200 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_precision_vs_oracle.py -q -m gpu -x --timeout 900 > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -n 5 $OUT/pytest.log
for rep in 1 2; do for v in 0 1; do
  NOMAD_BF16_P9=$v timeout 600 python bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $OUT/bench_c5_p9_${v}_$rep.json 2> $OUT/bench_c5_${v}_$rep.err
  echo "P9=$v rep $rep: $(python3 -c "import json,sys; d=json.load(open('$OUT/bench_c5_p9_${v}_$rep.json')); print(d['value'], d['ms_per_step'])")" | tee -a $OUT/summary.txt
done; done
