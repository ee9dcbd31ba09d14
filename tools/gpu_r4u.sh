#!/bin/bash
TAG=${1:-r4u}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do for f in 0.5 0.4375 0.5625 0.625 0.375; do
  NOMAD_F32_SPLIT_FRAC=$f timeout 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-profile --live-traffic off > $OUT/b.json 2> $OUT/bench.err
  python3 -c "
import json
d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1])
print('frac $f', d['value'], d['ms_per_step'])
"
done; done | tee $OUT/frac.txt
