#!/bin/bash
python -m pytest tests/test_gpu_gemm_x3.py tests/test_gpu_race_screen.py -x -q -m gpu 2>&1 | tail -3
python tools/bench_c4.py --precision bf16x3 2>/dev/null | tail -1
python tools/bench_train.py --gemm-precision bf16x3 --steps 5 2>/dev/null | tail -1 | cut -c1-330
python tools/bench_small_batch.py 2>/dev/null | grep '"samples": 64000' | cut -c1-150 | head -5
