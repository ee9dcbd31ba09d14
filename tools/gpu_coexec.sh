#!/bin/bash
# do MFMA and VALU streams of different waves on one SIMD overlap? (tools/micro/coexec.hip)
TAG=${1:-coexec}
OUT=gpurun_out/$TAG; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -o /tmp/coexec tools/micro/coexec.hip > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
timeout 120 /tmp/coexec 2>&1 | tee $OUT/coexec.txt
