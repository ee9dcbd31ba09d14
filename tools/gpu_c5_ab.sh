#!/bin/bash
# C5 per-kernel times, shipped build vs the A/B build with packed FP32, alternating on one box
for i in 1 2; do
for v in shipped pk; do
  if [ $v = pk ]; then export NOMAD_LIB_VARIANT=pk; else unset NOMAD_LIB_VARIANT; fi
  python bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', d['value'], 'clips/s', d['kernel_time_ms_per_step'])"
done; done
