#!/bin/bash
# round 2, first GPU trip: the new parity tests (feature_grad_mult, RCCL at world size 1, bf16 / bf16x3 vs the oracle), bench
TAG=${1:-r02a}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1700 python -m pytest tests/test_gpu_backward.py tests/test_gpu_rccl.py tests/test_gpu_precision_vs_oracle.py tests/test_gpu_bf16.py tests/test_gpu_bench_contract.py -q -m gpu -s --timeout 900 > $OUT/pytest_new.log 2>&1; echo "pytest(new) exit $?" | tee -a $OUT/summary.txt
grep -E "vs oracle|rel err|passed|failed|error|Error" $OUT/pytest_new.log | tail -n 40
timeout 900 python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" | tee -a $OUT/summary.txt
cat $OUT/bench.json; tail -n 3 $OUT/bench.err
