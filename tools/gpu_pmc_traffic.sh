#!/bin/bash
# HBM-side traffic of the bench kernels: two separate PMC passes of bench.py, then tools/pmc_traffic.py.
# Usage: bash tools/gpu_pmc_traffic.sh <tag>
TAG=${1:-r01t}
OUT=gpurun_out/$TAG; mkdir -p $OUT
ROOTDIR=$(pwd)
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $ROOTDIR/$OUT/pmc_$c -o p -- python3 $ROOTDIR/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-also > $ROOTDIR/$OUT/pmc_$c.log 2>&1
  echo "pmc $c exit $?" | tee -a $ROOTDIR/$OUT/summary.txt
done
cd $ROOTDIR
mkdir -p $OUT/profiles_out
python3 tools/pmc_traffic.py $OUT $TAG | head -30
cp profiles/${TAG}_pmc_traffic.json $OUT/
find $OUT -name "*.csv" -size +6M -delete
