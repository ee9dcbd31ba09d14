#!/bin/bash
TAG=${1:-bf16}
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_bf16.py -q -m gpu -s --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -n 30 $OUT/pytest.log | cut -c1-300
timeout 600 python3 tools/gemm_sweep.py --bf16 --tiles 0,1,3,5,6 --shapes qkv,fc1,fc2,out,conv3 --iters 7 --json $OUT/sweep_bf16.json 2>&1 | grep -v amdgpu
