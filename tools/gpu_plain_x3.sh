#!/bin/bash
TAG=${1:-plainx3}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_gemm_x3.py tests/test_gpu_bf16x3.py tests/test_gpu_train.py -q -m gpu -x > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -n 2 $OUT/pytest.log
for rep in 1 2; do for m in 0 1; do
  export NOMAD_F32_PLAIN_EPI=$m
  echo -n "plain_epi=$m c4 x3: "; timeout 300 python3 tools/bench_c4.py --precision bf16x3 2>>$OUT/err.log | tail -1 | cut -c1-260
  echo -n "plain_epi=$m train x3: "; timeout 300 python3 tools/bench_train.py --gemm-precision bf16x3 --steps 6 2>>$OUT/err.log | tail -1 | cut -c150-330
done; done
