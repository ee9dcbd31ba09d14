#!/bin/bash
# round 5, trip sk: split-K epilogue + LayerNorm in one kernel (configs[3]) - bit-identity test, backward / parity tests, configs[3] timing A/B
TAG=${1:-s5sk}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_backward.py tests/test_gpu_parity.py tests/test_reference_classes.py tests/test_gpu_race_screen.py -q -m gpu -x --timeout 900 > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -n 5 $OUT/pytest.log
for rep in 1 2 3; do for v in 0 1; do
  NOMAD_DIAG_LIB=1 NOMAD_SPLITK_LN=$v timeout 300 python3 tools/c4_profile.py 200 > $OUT/c4_${v}_$rep.txt 2>&1
  echo "splitk_ln=$v rep $rep: $(grep forward+backward $OUT/c4_${v}_$rep.txt)" | tee -a $OUT/summary.txt
done; done
