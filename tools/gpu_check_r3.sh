#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/ -x -q -m gpu > gpurun_out/r3_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/r3_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3_bench.json"))
print(d["value"], d["roofline"]["frac"], d["also_measured"]["value"], d["also_measured_c5"]["value"], d["also_measured_c3"]["value"], d["also_measured_c4"])
PY
