#!/bin/bash
# round 3, final trip: the driver's GPU tier (tests, smoke, bench) + the rocprofv3 summaries committed under profiles/
mkdir -p gpurun_out/r3prof
python -m pytest tests/ -x -q -m gpu > gpurun_out/r3_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/r3_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err; echo "bench rc=$?"
ROOTDIR=$(pwd); export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/gpurun_out/r3prof/c2 -o c2 -- python3 $ROOTDIR/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --single-stream --no-also > $ROOTDIR/gpurun_out/r3prof/c2.log 2>&1; echo "c2 stats rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/gpurun_out/r3prof/c5 -o c5 -- python3 $ROOTDIR/bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 5 --warmup 2 --no-cpu-baseline --no-profile --single-stream > $ROOTDIR/gpurun_out/r3prof/c5.log 2>&1; echo "c5 stats rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/gpurun_out/r3prof/c4x3 -o c4x3 -- python3 $ROOTDIR/tools/bench_c4.py --precision bf16x3 > $ROOTDIR/gpurun_out/r3prof/c4x3.log 2>&1; echo "c4 stats rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/gpurun_out/r3prof/trainx3 -o trainx3 -- python3 $ROOTDIR/tools/bench_train.py --gemm-precision bf16x3 --steps 4 > $ROOTDIR/gpurun_out/r3prof/trainx3.log 2>&1; echo "train stats rc=$?"
cd $ROOTDIR
find gpurun_out/r3prof -name "*_kernel_trace.csv" -delete; find gpurun_out/r3prof -name "*agent_info.csv" -delete
tail -1 gpurun_out/r3prof/c4x3.log | cut -c1-200; tail -1 gpurun_out/r3prof/trainx3.log | cut -c1-300
