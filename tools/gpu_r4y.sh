#!/bin/bash
# ablations of the pipelined fp32 attention kernel (timing probes)
TAG=${1:-r4y}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for sc in 0.5 0.05; do
for a in 0 1 2 16 18 19 23; do
  SCALE=$sc DIAG=1 NOMAD_ATTN_ABLATE=$a timeout 120 python3 tools/attn_one.py 2>/dev/null
done
SCALE=$sc DIAG=1 NOMAD_ATTN_PIPE=0 timeout 120 python3 tools/attn_one.py 2>/dev/null
done | tee $OUT/ablate.txt
