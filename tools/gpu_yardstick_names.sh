#!/bin/bash
# which vendor kernels (Tensile macro-tile / MFMA shape in the kernel name) torch.matmul picks on the hot shapes, with their durations
TAG=${1:-yard_names}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o y -- python3 $ROOTDIR/tools/lib_gemm_yardstick.py --iters 4 > $OUT/yardstick.jsonl 2> $OUT/err.log; echo "exit $?"
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv && cut -c1-260 $OUT/kernel_stats.csv | head -40
find $OUT/prof -type f ! -name "*stats*" -delete 2>/dev/null
