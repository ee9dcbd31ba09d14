#!/bin/bash
# round 5, trip b: persistent bf16 GEMM per shape (A/B vs shipped + vendor), its per-workgroup timeline, the old kernel's timeline
TAG=${1:-s5b}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python3 tools/p9_ab.py > $OUT/p9_ab.jsonl 2> $OUT/p9_ab.err; echo "p9_ab exit $?" | tee -a $OUT/summary.txt
cat $OUT/p9_ab.jsonl; tail -3 $OUT/p9_ab.err
timeout 600 python3 tools/p9_timeline.py > $OUT/p9_timeline.jsonl 2> $OUT/p9_timeline.err; echo "p9_timeline exit $?" | tee -a $OUT/summary.txt
cat $OUT/p9_timeline.jsonl; tail -3 $OUT/p9_timeline.err
timeout 600 python3 tools/gemm_timeline.py --shapes c5_qkv,c5_fc1,c5_fc2,c5_out > $OUT/p8_timeline.jsonl 2> $OUT/p8_timeline.err; echo "p8_timeline exit $?" | tee -a $OUT/summary.txt
cat $OUT/p8_timeline.jsonl
