#!/bin/bash
# A/B of the bf16 attention kernels on the GPU box.  Usage: bash tools/gpu_attn_micro.sh <tag>
TAG=${1:-attn}
OUT=gpurun_out/$TAG; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -o /tmp/attn_bf16 tools/micro/attn_bf16.hip > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
for cfg in "32 1499 3"; do
  timeout 300 /tmp/attn_bf16 $cfg 2>&1 | tee -a $OUT/attn_micro.txt
done
