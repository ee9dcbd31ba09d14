#!/bin/bash
# round 3: GPU test tier again + the rocprofv3 summaries committed under profiles/ (kernel stats of the bench commands, PMC traffic)
mkdir -p gpurun_out/r3prof
python -m pytest tests/ -x -q -m gpu > gpurun_out/r3_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 4 gpurun_out/r3_pytest_gpu.log
ROOTDIR=$(pwd); export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $ROOTDIR/gpurun_out/r3prof/c2 -o c2 -- python3 $ROOTDIR/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --single-stream --no-also > $ROOTDIR/gpurun_out/r3prof/c2.log 2>&1; echo "c2 stats rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats -d $ROOTDIR/gpurun_out/r3prof/c5 -o c5 -- python3 $ROOTDIR/bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 5 --warmup 2 --no-cpu-baseline --no-profile --single-stream > $ROOTDIR/gpurun_out/r3prof/c5.log 2>&1; echo "c5 stats rc=$?"
cd $ROOTDIR
find gpurun_out/r3prof -name "*kernel_stats.csv" | while read f; do cp $f gpurun_out/r3prof/$(basename $(dirname $(dirname $f)))_$(basename $f); done
find gpurun_out/r3prof -name "*_kernel_trace.csv" -delete; find gpurun_out/r3prof -name "*.db" -delete
ls -la gpurun_out/r3prof | head -20
bash tools/gpu_pmc_traffic.sh r03 2>&1 | tail -8
