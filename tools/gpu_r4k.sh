#!/bin/bash
# round 4, trip 11: the new variants wired into the forward: parity tests, bench, env A/B
TAG=${1:-r4k}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py tests/test_reference_classes.py -q -m gpu -x --timeout 900 > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -n 5 $OUT/pytest.log
for cfg in "1 1 1 1" "0 0 0 0" "1 1 1 1" "1 0 1 1" "1 1 0 1" "1 1 1 0"; do
  set -- $cfg
  NOMAD_F32_LEAN=$1 NOMAD_F32_DIRECT_EPI=$2 NOMAD_F32_SKEW=$3 NOMAD_F32_QUANT_TILE=$4 timeout 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --live-traffic off > $OUT/bench_$1$2$3$4.json 2> $OUT/bench.err
  python3 -c "
import json,sys
d=json.loads(open('$OUT/bench_$1$2$3$4.json').read().strip().splitlines()[-1])
print('lean/direct/skew/quant $1$2$3$4', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['kernel_time_ms_per_step'])
"
done
