#!/bin/bash
python -m pytest tests/test_gpu_gemm_x3.py tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -3
python tools/bench_train.py --gemm-precision bf16x3 --steps 5 2>/dev/null | tail -1 | cut -c1-900
python tools/bench_train.py --gemm-precision fp32 --steps 5 2>/dev/null | tail -1 | cut -c1-900
