#!/bin/bash
# round 4, trip 1: loop-skeleton micro, A/B of the new fp32 GEMM instantiations against production and the vendor, PMC of both
TAG=${1:-r4a}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_loop2 tools/micro/mfma_loop2.hip 2> $OUT/micro_build.log && timeout 300 /tmp/mfma_loop2 > $OUT/micro.txt 2>&1
echo "micro exit $?" | tee -a $OUT/summary.txt
timeout 900 python3 tools/gemm_ab.py --tiles 33,60,62,65,66,67,31,61,63,-1 --shapes qkv,out,fc1,fc2,conv3,fc2_2r > $OUT/gemm_ab.jsonl 2> $OUT/gemm_ab.err
echo "gemm_ab exit $?" | tee -a $OUT/summary.txt
timeout 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --live-traffic off > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?" | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x --timeout 600 > $OUT/pytest_kernels.log 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -n 3 $OUT/pytest_kernels.log
for sh in fc2 conv3; do bash tools/gpu_pmc_vendor.sh $TAG/pmc 33,65,-1 $sh > $OUT/pmc_$sh.log 2>&1; done
cat $OUT/micro.txt; cat $OUT/gemm_ab.jsonl; tail -c 1500 $OUT/bench.json
