#!/bin/bash
# round 3, hunt 1: does the bf16 nondeterminism reproduce (product / diag library), and which stage differs first?
mkdir -p gpurun_out
export HUNT_PARTS=a,b,c,d
( timeout 300 python tools/race_hunt_stages.py 80 ) > gpurun_out/h1_product.log 2>&1
( HUNT_DIAG=1 timeout 300 python tools/race_hunt_stages.py 80 ) > gpurun_out/h1_diag.log 2>&1
( HUNT_CKSUM=1 timeout 900 python tools/race_hunt_stages.py 300 ) > gpurun_out/h1_cksum.log 2>&1
tail -n 30 gpurun_out/h1_product.log gpurun_out/h1_diag.log gpurun_out/h1_cksum.log
