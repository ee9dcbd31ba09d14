#!/bin/bash
# 256x192 tiles of the deep-pipelined bf16 GEMM: parity tests, isolated A/B (yardstick), C5 bench A/B (NOMAD_BF16_N192=0 / auto)
TAG=${1:-n192}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16.py -q -m gpu -x --timeout 600 -k "gemm" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -n 3 $OUT/pytest.log
timeout 300 python3 tools/lib_gemm_yardstick.py --f32 "" > $OUT/yardstick.jsonl 2> $OUT/err.log; echo "yard exit $?"
cat $OUT/yardstick.jsonl
for rep in 1 2; do
for m in 0 auto; do
  if [ $m = auto ]; then unset NOMAD_BF16_N192; else export NOMAD_BF16_N192=$m; fi
  timeout 300 python3 bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/c5_$m.$rep.json 2> $OUT/c5_$m.$rep.err; echo "c5 N192=$m rep $rep exit $?"
  python3 -c "import json,sys; d=json.loads(open('$OUT/c5_$m.$rep.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('kernel_time_ms_per_step'))"
done; done
