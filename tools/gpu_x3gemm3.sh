#!/bin/bash
mkdir -p gpurun_out/r3prof
python -m pytest tests/test_gpu_gemm_x3.py tests/test_gpu_bf16x3.py -x -q -m gpu 2>&1 | tail -4
for p in fp32 bf16x3; do python tools/bench_c4.py --precision $p 2>/dev/null | tail -1; done
python tools/bench_train.py --gemm-precision bf16x3 --steps 5 2>/dev/null | tail -1 | cut -c1-420
python tools/bench_small_batch.py 2>/dev/null | grep '"samples": 64000' | cut -c1-200
ROOTDIR=$(pwd); export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/gpurun_out/r3prof/c4x3 -o c4x3 -- python3 $ROOTDIR/tools/bench_c4.py --precision bf16x3 > $ROOTDIR/gpurun_out/r3prof/c4x3.log 2>&1; echo "c4 stats rc=$?"
cd $ROOTDIR; find gpurun_out/r3prof -name "*_kernel_trace.csv" -delete; find gpurun_out/r3prof -name "*agent_info.csv" -delete
