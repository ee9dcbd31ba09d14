#!/bin/bash
TAG=${1:-r6f}
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
  run NOMAD_F32_SPLIT_WAYS=2
  run NOMAD_F32_SPLIT_WAYS=1
  run NOMAD_F32_SPLIT_WAYS=3
  run NOMAD_F32_SPLIT_WAYS=2 NOMAD_F32_QUANT_PENALTY=0
  run NOMAD_F32_SPLIT_WAYS=2 NOMAD_F32_QUANT_PENALTY=6
done | tee $OUT/ab.txt
