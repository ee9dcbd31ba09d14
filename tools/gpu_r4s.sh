#!/bin/bash
# round 4: PMC passes on the shipped fp32 GEMM instantiations (tile 90 = lean + skewed + direct epilogue, 91 = lean + skewed, LDS epilogue) and the vendor kernel
TAG=${1:-r4s}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
bash tools/gpu_pmc_vendor.sh $TAG 90,-1 qkv > $OUT/pmc_qkv.log 2>&1
bash tools/gpu_pmc_vendor.sh $TAG 91,-1 fc2 > $OUT/pmc_fc2.log 2>&1
bash tools/gpu_pmc_vendor.sh $TAG 90,-1 conv3 > $OUT/pmc_conv3.log 2>&1
for s in qkv fc2 conv3; do echo "##### $s"; cat $OUT/$s/pmc_summary.txt; done
