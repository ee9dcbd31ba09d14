#!/bin/bash
# round 4, trip 13: headline bench, two-stream split on / off, alternating on one box
TAG=${1:-r4m}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do for ss in "" "--single-stream"; do
  timeout 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-profile --live-traffic off $ss > $OUT/b.json 2> $OUT/bench.err
  python3 -c "
import json
d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1])
print('split' if '$ss'=='' else 'single', d['value'], d['ms_per_step'])
"
done; done
