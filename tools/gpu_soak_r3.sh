#!/bin/bash
# round 3 soak: the ORIGINAL round-2 reproducer (two-stream bf16 split, fp32 embeds in between, prior activity a-d) and the
# whole-forward victims, on the shipped build (no packed FP32): every count must be 0
mkdir -p gpurun_out
( HUNT_PARTS=a,b,c,d timeout 900 python tools/race_hunt_stages.py 1500 ) > gpurun_out/r3_soak_original_reproducer.log 2>&1
grep "^lib=\|differs" gpurun_out/r3_soak_original_reproducer.log | tail -5
( timeout 900 python tools/race_hunt_forward.py 30 bf16,fp32,bf16x3 bt1,bt11,bt12,forward16,x3 ) > gpurun_out/r3_soak_forward.log 2>&1
grep "^library" gpurun_out/r3_soak_forward.log
( HUNT_PROVENANCE=0 timeout 300 python tools/race_hunt_conv0.py 40 bt11,bt12,forward ) > gpurun_out/r3_soak_conv0.log 2>&1
grep "^variant" gpurun_out/r3_soak_conv0.log
