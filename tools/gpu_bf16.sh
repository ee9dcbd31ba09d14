#!/bin/bash
TAG=${1:-bf16}
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_bf16.py -q -m gpu -s --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -n 30 $OUT/pytest.log | cut -c1-300
python bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 5 --warmup 2 --no-cpu-baseline | tee $OUT/bench_c5_bf16.json
python bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline | tee $OUT/bench_c2_bf16.json
