#!/bin/bash
# headline (configs[1]) A/B of one environment switch, alternating.  Usage: bash tools/gpu_env_ab_c2.sh <tag> <ENV_NAME> <value A> <value B>
TAG=${1:-envabc2}; VAR=$2; A=$3; B=$4
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2 3; do for m in $A $B; do
  export $VAR=$m
  timeout 300 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-also --live-traffic off > $OUT/c2_$m.$rep.json 2> $OUT/c2_$m.$rep.err; echo -n "c2 $VAR=$m rep $rep exit $?  "
  python3 -c "import json,sys; d=json.loads(open('$OUT/c2_$m.$rep.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('kernel_time_ms_per_step'))"
done; done
