#!/bin/bash
TAG=${1:-clockc5}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for mode in single split; do for m in 0 auto 0 auto; do
  if [ $m = auto ]; then unset NOMAD_BF16_N192; else export NOMAD_BF16_N192=$m; fi
  timeout 200 python3 tools/clock_c5.py $mode 2>> $OUT/err.log | tee -a $OUT/clock.jsonl
done; done
