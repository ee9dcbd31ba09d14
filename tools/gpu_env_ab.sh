#!/bin/bash
# C5 bench A/B of one environment switch, alternating.  Usage: bash tools/gpu_env_ab.sh <tag> <ENV_NAME> <value A> <value B>
TAG=${1:-envab}; VAR=$2; A=$3; B=$4
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2 3; do for m in $A $B; do
  export $VAR=$m
  timeout 300 python3 bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/c5_$m.$rep.json 2> $OUT/c5_$m.$rep.err; echo -n "c5 $VAR=$m rep $rep exit $?  "
  python3 -c "import json,sys; d=json.loads(open('$OUT/c5_$m.$rep.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('kernel_time_ms_per_step'))"
done; done
