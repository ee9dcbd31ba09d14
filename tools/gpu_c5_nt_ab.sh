#!/bin/bash
# same-box alternating A/B of an environment switch of the bf16 GEMM on config C5 end to end
# Usage: bash tools/gpu_c5_nt_ab.sh [VAR]      (NOMAD_BF16_NT_STORES, NOMAD_BF16_B3)
VAR=${1:-NOMAD_BF16_B3}
for i in 1 2 3; do for v in 0 1; do
  env $VAR=$v python bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 10 --warmup 3 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5 $VAR=$v', d['value'], d['ms_per_step'])"
done; done
