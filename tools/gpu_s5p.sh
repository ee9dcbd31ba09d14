#!/bin/bash
# round 5, trip p: race screen of the persistent bf16 GEMM
TAG=${1:-s5p}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python3 tools/soak_p9.py 60 > $OUT/soak_p9.txt 2> $OUT/soak_p9.err; echo "soak exit $?" | tee -a $OUT/summary.txt
cat $OUT/soak_p9.txt; tail -3 $OUT/soak_p9.err
