#!/bin/bash
# round 3, hunt 2: conv0 alone against one class of co-running kernels; then the full loop with copies of conv0's inputs / output
mkdir -p gpurun_out
( timeout 400 python tools/race_hunt_conv0.py 15 ) > gpurun_out/h2_conv0_micro.log 2>&1
( HUNT_PARTS=d timeout 300 python tools/race_hunt_conv0.py 15 forward,gemm16,gemm3,attn ) > gpurun_out/h2_conv0_micro_parts.log 2>&1
( HUNT_PARTS=a,b,c,d HUNT_CKSUM=1 HUNT_SNAP=1 timeout 800 python tools/race_hunt_stages.py 1500 ) > gpurun_out/h2_snap.log 2>&1
tail -n 40 gpurun_out/h2_conv0_micro.log gpurun_out/h2_conv0_micro_parts.log gpurun_out/h2_snap.log
