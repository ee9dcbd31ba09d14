#!/bin/bash
# LDS bank-conflict counters of the bf16x3 GEMM kernels (QKV shape).  Usage: bash tools/gpu_pmc_lds.sh <tag> "<variants>"
TAG=${1:-pmclds}; VARS=${2:-"8 1"}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for v in $VARS; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/v$v -o p -- python3 $ROOTDIR/tools/gemm_x3_one.py qkv $v > $OUT/v$v.log 2>&1
  echo "variant $v exit $?" >> $OUT/summary.txt
done
cd $ROOTDIR
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/v*/")):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_bf16" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    line = d.rstrip("/").split("/")[-1] + ": " + ", ".join(f"{k}={sum(v)/len(v):.4g}" for k, v in sorted(agg.items()))
    print(line)
    open("$OUT/lds_summary.txt", "a").write(line + "\n")
PY
find $OUT -name "*.csv" -size +4M -delete
