#!/bin/bash
# round 5, trip p3: the slab pos-conv - its own test, the kernel alone against the grouped GEMM, then the full GPU suite
TAG=${1:-s5p3}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16.py -q -m gpu -x --timeout 900 -k posconv > $OUT/pytest_posconv.log 2>&1; echo "pytest posconv exit $?" | tee -a $OUT/summary.txt
tail -n 12 $OUT/pytest_posconv.log
timeout 600 python3 tools/posconv_time.py > $OUT/posconv_time.jsonl 2> $OUT/posconv_time.err; cat $OUT/posconv_time.jsonl; tail -2 $OUT/posconv_time.err
timeout 2700 python -m pytest tests -q -m gpu -x --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -n 4 $OUT/pytest_gpu.log
