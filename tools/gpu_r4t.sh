#!/bin/bash
# round 4: how many stream-parallel parts should the fp32 bench batch run as, with the round-4 kernels?
TAG=${1:-r4t}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do for w in 1 2 3 4; do
  NOMAD_F32_SPLIT_WAYS=$w timeout 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-profile --live-traffic off > $OUT/b.json 2> $OUT/bench.err
  python3 -c "
import json
d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1])
print('ways $w', d['value'], d['ms_per_step'])
"
done; done | tee $OUT/ways.txt
