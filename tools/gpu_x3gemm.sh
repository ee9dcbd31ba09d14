#!/bin/bash
# round 3: the bf16x3-products mode of the fp32-layout GEMMs - tests, C4 / training timings, then the profile summaries again
mkdir -p gpurun_out/r3prof
python -m pytest tests/test_gpu_gemm_x3.py tests/test_gpu_bench_contract.py -x -q -m gpu 2>&1 | tail -15
for p in fp32 bf16x3; do python tools/bench_c4.py --precision $p 2>/dev/null | tail -1; done
python tools/bench_c4.py --precision bf16x3 --batch 8 2>/dev/null | tail -1
python tools/bench_c4.py --precision fp32 --batch 8 2>/dev/null | tail -1
ROOTDIR=$(pwd); export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/gpurun_out/r3prof/c2 -o c2 -- python3 $ROOTDIR/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --single-stream --no-also > $ROOTDIR/gpurun_out/r3prof/c2.log 2>&1; echo "c2 stats rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/gpurun_out/r3prof/c5 -o c5 -- python3 $ROOTDIR/bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 5 --warmup 2 --no-cpu-baseline --no-profile --single-stream > $ROOTDIR/gpurun_out/r3prof/c5.log 2>&1; echo "c5 stats rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/gpurun_out/r3prof/c4x3 -o c4x3 -- python3 $ROOTDIR/tools/bench_c4.py --precision bf16x3 > $ROOTDIR/gpurun_out/r3prof/c4x3.log 2>&1; echo "c4 stats rc=$?"
cd $ROOTDIR
find gpurun_out/r3prof -name "*_kernel_trace.csv" -delete; find gpurun_out/r3prof -name "*agent_info.csv" -delete
find gpurun_out/r3prof -type f | head -20
bash tools/gpu_pmc_traffic.sh r03 2>&1 | tail -4
