#!/bin/bash
# C5 forward time + held shader clock for two settings of one environment switch, alternating (tools/clock_c5.py)
# Usage: bash tools/gpu_clock_env.sh <tag> <ENV_NAME> <value A> <value B>
TAG=${1:-clockenv}; VAR=$2; A=$3; B=$4
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for mode in single split; do for rep in 1 2; do for m in $A $B; do
  export $VAR=$m
  echo -n "$VAR=$m " | tee -a $OUT/clock.jsonl
  timeout 200 python3 tools/clock_c5.py $mode 2>> $OUT/err.log | tee -a $OUT/clock.jsonl
done; done; done
