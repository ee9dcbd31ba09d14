#!/bin/bash
TAG=${1:-r6c}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -q -x -k "attention" 2>&1 | tail -4
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_backward.py -q -x 2>&1 | tail -3
for sc in 0.5 0.05; do SCALE=$sc timeout 120 python3 tools/attn_one.py 2>/dev/null; done | tee $OUT/attn.txt
T=150 timeout 120 python3 tools/attn_one.py 2>/dev/null; T=400 B=64 timeout 120 python3 tools/attn_one.py 2>/dev/null; T=1499 B=8 timeout 120 python3 tools/attn_one.py 2>/dev/null
timeout 200 python3 tools/clock_under_load.py 2>/dev/null | tail -1
for i in 1 2; do timeout 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-profile --live-traffic off 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])
"; done
