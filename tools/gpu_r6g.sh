#!/bin/bash
TAG=${1:-r6g}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
run() {
  env "$@" timeout 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-profile --live-traffic off > $OUT/b.json 2> $OUT/bench.err
  python3 -c "
import json
d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1])
print('$*', d['value'], d['ms_per_step'])
"
}
for rep in 1 2 3; do
  for q in 3 6 8 10 15 25; do run NOMAD_F32_QUANT_PENALTY=$q; done
done | tee $OUT/ab.txt
