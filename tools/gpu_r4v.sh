#!/bin/bash
TAG=${1:-r4v}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do for q in 3 -4 -8 -15; do
  NOMAD_F32_QUANT_PENALTY=$q timeout 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-profile --live-traffic off > $OUT/b.json 2> $OUT/bench.err
  python3 -c "
import json
d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1])
print('penalty $q %', d['value'], d['ms_per_step'])
"
done; done | tee $OUT/penalty.txt
