#!/bin/bash
# the whole GPU parity suite while another PROCESS keeps the chip busy with the kernels that exposed the packed-FP32 hazard
mkdir -p gpurun_out
python tools/aggressor.py 900 > gpurun_out/r3_aggressor.log 2>&1 &
AG=$!
sleep 25
python -m pytest tests/ -q -m gpu -p no:cacheprovider > gpurun_out/r3_pytest_gpu_under_load.log 2>&1; echo "pytest under load rc=$?"
tail -n 12 gpurun_out/r3_pytest_gpu_under_load.log
kill $AG 2>/dev/null; wait $AG 2>/dev/null
tail -n 2 gpurun_out/r3_aggressor.log
