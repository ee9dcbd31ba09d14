#!/bin/bash
# plain-epilogue fp32 GEMM instantiations on the small-tile / training paths: tests, then C4 and fine-tuning step A/B
TAG=${1:-plaintrain}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_backward.py tests/test_gpu_kernels.py tests/test_gpu_gemm_x3.py tests/test_gpu_parity.py -q -m gpu -x > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -n 2 $OUT/pytest.log
for rep in 1 2; do for m in 0 1; do
  export NOMAD_F32_PLAIN_EPI=$m
  echo -n "plain_epi=$m c4: "; timeout 300 python3 tools/bench_c4.py 2>>$OUT/err.log | tail -1 | cut -c1-260
  echo -n "plain_epi=$m train: "; timeout 300 python3 tools/bench_train.py --steps 6 2>>$OUT/err.log | tail -1 | cut -c1-260
  echo -n "plain_epi=$m small batch: "; timeout 300 python3 tools/bench_small_batch.py 2>>$OUT/err.log | tail -2 | cut -c1-300
done; done
