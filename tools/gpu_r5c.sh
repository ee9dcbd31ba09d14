#!/bin/bash
# mixed tile shapes incl. the conv stack: bit check, parity tests, bench A/B over the split rule's range
TAG=${1:-r5c}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python3 tools/dbg_tr.py 99 2>&1 | grep -v "amdgpu.ids\|skipped" | cut -c1-160
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py tests/test_gpu_race_screen.py -q -x 2>&1 | tail -3
run() {
  env "$@" timeout 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-profile --live-traffic off > $OUT/b.json 2> $OUT/bench.err
  python3 -c "
import json
d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1])
print('$*', d['value'], d['ms_per_step'])
"
}
for rep in 1 2; do
  run NOMAD_F32_MIXED=1
  run NOMAD_F32_MIXED=0
  run NOMAD_F32_MIXED_MAX=0.5
  run NOMAD_F32_MIXED_MAX=0.85
  run NOMAD_F32_MIXED_MIN=0.15
  run NOMAD_F32_MIXED_SLOTS=256
  run NOMAD_F32_MIXED_SLOTS=256 NOMAD_F32_MIXED_MAX=0.85
done | tee $OUT/ab_bench.txt
