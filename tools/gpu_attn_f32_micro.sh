#!/bin/bash
# A/B of the fp32 attention kernels on the GPU box.  Usage: bash tools/gpu_attn_f32_micro.sh <tag>
TAG=${1:-attnf32}
OUT=gpurun_out/$TAG; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -o /tmp/attn_f32 tools/micro/attn_f32.hip > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
for cfg in "256 199 4" "32 50 2" "4 700 2" "3 257 2" "2 33 1"; do
  timeout 300 /tmp/attn_f32 $cfg 2>&1 | tee -a $OUT/attn_f32_micro.txt
done
