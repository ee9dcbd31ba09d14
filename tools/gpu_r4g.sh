#!/bin/bash
# round 4, trip 7: kernel arguments in device memory (HIP_FORCE_DEV_KERNARG=1) - does the per-workgroup set-up get shorter?
TAG=${1:-r4g}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for v in 0 1; do
  export HIP_FORCE_DEV_KERNARG=$v
  timeout 600 python3 tools/gemm_timeline_f32.py --shapes qkv --tiles 86,68 > $OUT/timeline_devkernarg$v.jsonl 2> $OUT/timeline.err
  timeout 900 python3 tools/gemm_ab.py --tiles 33,72,84,-1 --shapes qkv,fc1,conv3 > $OUT/gemm_ab_devkernarg$v.jsonl 2> $OUT/gemm_ab.err
done
for v in 0 1; do echo "== HIP_FORCE_DEV_KERNARG=$v"; cat $OUT/timeline_devkernarg$v.jsonl | cut -c1-900; cat $OUT/gemm_ab_devkernarg$v.jsonl | cut -c1-140; done
