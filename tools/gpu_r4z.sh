#!/bin/bash
TAG=${1:-r4z}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -q -x -k "attention or ragged" 2>&1 | tail -5
for sc in 0.5 0.05; do
for a in 0 1 2 16 18 19 23; do
  SCALE=$sc DIAG=1 NOMAD_ATTN_ABLATE=$a timeout 120 python3 tools/attn_one.py 2>/dev/null
done
SCALE=$sc DIAG=1 NOMAD_ATTN_PIPE=0 timeout 120 python3 tools/attn_one.py 2>/dev/null
done | tee $OUT/ablate.txt

T=400 B=64 timeout 120 python3 tools/attn_one.py 2>/dev/null; T=400 B=64 NOMAD_ATTN_PIPE=0 timeout 120 python3 tools/attn_one.py 2>/dev/null
T=1499 B=8 timeout 120 python3 tools/attn_one.py 2>/dev/null; T=1499 B=8 NOMAD_ATTN_PIPE=0 timeout 120 python3 tools/attn_one.py 2>/dev/null
NOMAD_ATTN_ABLATE=32 python3 tools/attn_timeline.py 2>/dev/null | head -3
