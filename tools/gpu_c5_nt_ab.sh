#!/bin/bash
# same-box alternating A/B of NOMAD_BF16_NT_STORES: config C5 (bf16) and the bf16x3 line of the headline config
for i in 1 2 3; do for v in 0 1; do
  NOMAD_BF16_NT_STORES=$v python bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 10 --warmup 3 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5 nt=$v', d['value'], d['ms_per_step'])"
  NOMAD_BF16_NT_STORES=$v python bench.py --dtype bf16x3 --steps 6 --warmup 2 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16x3 nt=$v', d['value'], d['ms_per_step'])"
done; done
