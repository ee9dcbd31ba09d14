#!/bin/bash
# A/B on the bench: residual-ahead epilogue on / off, alternating
TAG=${1:-r4x}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2 3 4; do for ra in 1 0; do
  NOMAD_F32_RES_AHEAD=$ra timeout 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-profile --live-traffic off > $OUT/b.json 2> $OUT/bench.err
  python3 -c "
import json
d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1])
print('res_ahead $ra', d['value'], d['ms_per_step'])
"
done; done | tee $OUT/ab_bench.txt
