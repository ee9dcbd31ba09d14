#!/bin/bash
# round 3, hunt 4: what replaces conv0's accumulator (provenance), which aggressor variants trigger it, which victim variants survive
mkdir -p gpurun_out
( timeout 300 python tools/race_hunt_conv0.py 8 gemm1,bt7,bt8,bt5,bt11,bt12,bt0,bt6,bt4,ft31,ft20,ft37,ft34 ) > gpurun_out/h4_aggressors.log 2>&1
grep "^variant\|elem\|A(j0\|  x =\|  w =" gpurun_out/h4_aggressors.log | head -60
for v in 7 8 9; do
  ( HUNT_VARIANT=$v HUNT_PROVENANCE=0 timeout 120 python tools/race_hunt_conv0.py 10 forward,gemm1 ) > gpurun_out/h4_variant$v.log 2>&1
  grep "^variant\|differ" gpurun_out/h4_variant$v.log | head -8
done
