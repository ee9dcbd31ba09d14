#!/bin/bash
# configs[4] (bf16, 30 s clips): 256x192 tiles for the N = 768 GEMMs re-measured (NOMAD_BF16_N192), alternating
TAG=${1:-r5i}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
run() {
  env "$@" timeout 600 python3 bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-profile --live-traffic off > $OUT/b.json 2> $OUT/bench.err
  python3 -c "
import json
d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1])
print('$*', d['value'], d['ms_per_step'])
"
}
for rep in 1 2 3; do
  run NOMAD_BF16_N192=0
  run NOMAD_BF16_N192=1
  run NOMAD_BF16_N192=3
  run NOMAD_BF16_N192=4
done | tee $OUT/ab_c5.txt
