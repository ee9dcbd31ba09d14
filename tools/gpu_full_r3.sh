#!/bin/bash
# round 3: the whole GPU tier as the driver runs it - tests, smoke, bench
mkdir -p gpurun_out
python -m pytest tests/ -x -q -m gpu > gpurun_out/r3_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 15 gpurun_out/r3_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err; echo "bench rc=$?"; cut -c1-1500 gpurun_out/r3_bench.json
