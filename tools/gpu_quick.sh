#!/bin/bash
# quick regression trip: parity + precision + ragged tests, then the C5 bench with kernel stats
TAG=${1:-quick}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py tests/test_gpu_bf16x3.py tests/test_gpu_precision_vs_oracle.py tests/test_gpu_backward.py -q -m gpu -x --timeout 900 > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -n 4 $OUT/pytest.log
bash tools/gpu_c5.sh $TAG notests
