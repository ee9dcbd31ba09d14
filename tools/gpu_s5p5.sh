#!/bin/bash
# round 5, trip p5: slab pos-conv with fragment-ordered weights - test, time alone, PMC, bf16 tests, configs[4] bench
TAG=${1:-s5p5}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16.py -q -m gpu -x --timeout 900 > $OUT/pytest_bf16.log 2>&1; echo "pytest bf16 exit $?" | tee -a $OUT/summary.txt
tail -n 3 $OUT/pytest_bf16.log
timeout 600 python3 tools/posconv_time.py > $OUT/posconv_time.jsonl 2> $OUT/posconv_time.err; cat $OUT/posconv_time.jsonl
bash tools/gpu_pmc_posconv.sh $TAG/pmc > /dev/null 2>&1; cat $OUT/pmc/pmc_posconv.txt | grep -i "durations\|MFMA_BUSY\|TA_TA_BUSY\|WAIT_INST_ANY\|WAVE_CYCLES\|TCP_TOTAL\|TCP_TCC"
for rep in 1 2; do
  timeout 600 python bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $OUT/bench_c5_$rep.json 2> $OUT/bench_c5_$rep.err
  echo "c5 rep $rep: $(python3 -c "import json,sys; d=json.load(open('$OUT/bench_c5_$rep.json')); print(d['value'], d['ms_per_step'])")" | tee -a $OUT/summary.txt
done
