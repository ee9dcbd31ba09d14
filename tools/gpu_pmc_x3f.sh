#!/bin/bash
# PMC passes for the bf16x3-products GEMM on fp32 buffers (gemm_f32_glds_kernel<..., X3>): one rocprofv3 run per counter group.
# Usage: bash tools/gpu_pmc_x3f.sh <tag> [shape] [tile]     (shapes: tools/gemm_sweep.py SHAPES; tiles 37 / 20 / 31 / 33)
TAG=${1:-pmcx3f}; SHAPE=${2:-tm_fc1}; TILE=${3:-20}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp GEMM_X3=1
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o p -- python3 $ROOTDIR/tools/gemm_one.py $TILE $SHAPE > $OUT/p$i.log 2>&1
  echo "pass $i ($grp) exit $?" >> $OUT/summary.txt
done
cd $ROOTDIR
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
dur = []
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "gemm_f32_glds" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in sorted(glob.glob("$OUT/p1/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "gemm_f32_glds" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open("$OUT/pmc_summary.txt", "w") as o:
    o.write("shape $SHAPE tile $TILE (bf16x3 products on fp32 buffers), per launch; kernel duration under the profiler (us): %s\n" % ", ".join("%.1f" % d for d in dur))
    for k, v in agg.items():
        line = f"{k}: n={len(v)} mean={sum(v)/len(v):.6g}"
        print(line); o.write(line + "\n")
PY
find $OUT -name "*.csv" -size +4M -delete
