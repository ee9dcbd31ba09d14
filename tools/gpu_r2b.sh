#!/bin/bash
# round 2, second trip: full GPU suite on the two-library build, smoke, pairwise roofline, predict pipeline, bench
TAG=${1:-r02b}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/summary.txt; tail -3 $OUT/smoke.log
timeout 2400 python -m pytest tests -q -m gpu -x --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -n 6 $OUT/pytest_gpu.log
python tools/bench_pairwise.py > $OUT/pairwise.json 2> $OUT/pairwise.err; cat $OUT/pairwise.json
for p in fp32 bf16x3; do python tools/bench_predict.py --deg 9000 --ref 1000 --precision $p > $OUT/predict_$p.json 2> $OUT/predict_$p.err; tail -1 $OUT/predict_$p.json; done
timeout 900 python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" | tee -a $OUT/summary.txt
cut -c1-600 $OUT/bench.json
