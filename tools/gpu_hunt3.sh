#!/bin/bash
# round 3, hunt 3: which way of issuing conv0's two transcendental instructions survives a co-running bf16 forward / 128x128 GEMM?
mkdir -p gpurun_out
for v in 0 1 2 3 4 5 6; do
  ( HUNT_VARIANT=$v timeout 200 python tools/race_hunt_conv0.py 12 forward,gemm1 ) > gpurun_out/h3_variant$v.log 2>&1
  grep "^variant" gpurun_out/h3_variant$v.log
done
