#!/bin/bash
# C5 forward time + held clock per NOMAD_BF16_N192 mode (0 off, 1 where a round is saved, 3 fc2 only, 4 short-K only), alternating
TAG=${1:-clockc5}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for mode in single split; do for rep in 1 2; do for m in 0 1 3 4; do
  export NOMAD_BF16_N192=$m
  timeout 200 python3 tools/clock_c5.py $mode 2>> $OUT/err.log | tee -a $OUT/clock.jsonl
done; done; done
