#!/bin/bash
TAG=${1:-r5l}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_gpu_backward.py tests/test_gpu_race_screen.py tests/test_gpu_train.py tests/test_reference_classes.py -q -x -m gpu 2>&1 | tail -4
for rep in 1 2 3; do
  NOMAD_SPLITK_POSCONV=1 python3 tools/c4_profile.py 30 2>/dev/null | grep forward | sed 's/^/posconv split on:  /'
  NOMAD_SPLITK_POSCONV=0 python3 tools/c4_profile.py 30 2>/dev/null | grep forward | sed 's/^/posconv split off: /'
done | tee $OUT/c4_ab.txt
