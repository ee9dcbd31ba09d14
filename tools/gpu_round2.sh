#!/bin/bash
TAG=${1:-r01b}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -n 25 $OUT/pytest_gpu.log
rocprofv3 -L > $OUT/counters.txt 2>&1
timeout 900 python3 tools/gemm_sweep.py --json $OUT/sweep.json 2>&1 | tee $OUT/sweep.log
bash tools/gpu_pmc.sh $TAG/pmc_t0_fc1 0 fc1_nogelu
timeout 600 python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" | tee -a $OUT/summary.txt
cat $OUT/bench.json
