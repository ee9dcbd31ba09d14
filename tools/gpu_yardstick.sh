#!/bin/bash
# vendor-library yardstick on the hot GEMM shapes (C2 fp32, C5 bf16), alternating with the shipped kernels
TAG=${1:-yardstick}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python3 tools/lib_gemm_yardstick.py > $OUT/yardstick.jsonl 2> $OUT/err.log; echo "exit $?"
cat $OUT/yardstick.jsonl; tail -5 $OUT/err.log
