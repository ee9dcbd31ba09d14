#!/bin/bash
# round 4: config C4 A/B (split-K for the layers branch, lean set-up on the small tiles), batch-1 latency, tests of the loss path
TAG=${1:-r4r}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for cfg in "1 1" "0 0" "1 1" "0 1" "1 0"; do
  set -- $cfg
  echo "NOMAD_SPLITK_LAYERS=$1 NOMAD_F32_LEAN=$2: $(NOMAD_SPLITK_LAYERS=$1 NOMAD_F32_LEAN=$2 timeout 300 python3 tools/bench_c4.py 2>/dev/null | tail -1)"
done | tee $OUT/c4_ab.txt
timeout 1800 python -m pytest tests/test_gpu_backward.py tests/test_gpu_parity.py tests/test_reference_classes.py tests/test_gpu_kernels.py tests/test_gpu_abi_errors.py -q -m gpu -x --timeout 900 > $OUT/pytest.log 2>&1; echo "pytest exit $?"
tail -n 4 $OUT/pytest.log
