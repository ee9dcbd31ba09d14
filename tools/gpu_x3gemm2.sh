#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_gemm_x3.py tests/test_gpu_bf16x3.py -x -q -m gpu 2>&1 | tail -6
for p in fp32 bf16x3; do python tools/bench_c4.py --precision $p 2>/dev/null | tail -1; done
for p in fp32 bf16x3; do python tools/bench_train.py --gemm-precision $p --steps 5 2>/dev/null | tail -1 | cut -c1-700; done
