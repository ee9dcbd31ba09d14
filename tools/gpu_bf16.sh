#!/bin/bash
TAG=${1:-bf16}
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_bf16.py -q -m gpu -s --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -n 8 $OUT/pytest.log | cut -c1-300
timeout 600 python3 tools/gemm_sweep.py --bf16 --tiles 1,1,0,3,5,7 --shapes qkv,fc1,fc2,out,conv3 --iters 7 --json $OUT/sweep_bf16.json 2>&1 | grep -v amdgpu
python bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_c5_bf16.json; python bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_c2_bf16.json
python - <<PY
import json
for f in ("$OUT/bench_c5_bf16.json", "$OUT/bench_c2_bf16.json"):
    r = json.loads(open(f).read().strip().split("\n")[-1]); print(r["value"], r["ms_per_step"], r["kernel_time_ms_per_step"], r["roofline"]["achieved"])
PY
