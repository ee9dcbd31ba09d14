#!/bin/bash
# conv0 of the bf16 path on the matrix cores: parity tests, then C5 A/B (NOMAD_BF16_CONV0_MFMA=0 / 1), alternating, with kernel times
TAG=${1:-conv0mfma}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp

timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_precision_vs_oracle.py tests/test_gpu_race_screen.py -q -m gpu -x --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -n 4 $OUT/pytest.log
for rep in 1 2; do for m in 0 1; do
  export NOMAD_BF16_CONV0_MFMA=$m
  timeout 300 python3 bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/c5_$m.$rep.json 2> $OUT/c5_$m.$rep.err; echo "c5 conv0_mfma=$m rep $rep exit $?"
  python3 -c "import json,sys; d=json.loads(open('$OUT/c5_$m.$rep.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('kernel_time_ms_per_step'))"
done; done
