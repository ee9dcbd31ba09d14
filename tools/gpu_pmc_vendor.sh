#!/bin/bash
# PMC passes on the vendor's fp32 kernel and on nomad_diag_gemm tiles for one shape (one rocprofv3 run per counter group,
# kernel-trace only next to --pmc).  Usage: bash tools/gpu_pmc_vendor.sh <tag> <tiles e.g. 33,65,-1> <shape>
TAG=${1:-pmcv}; TILES=${2:-33,-1}; SHAPE=${3:-fc2}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG/$SHAPE; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o p -- python3 $ROOTDIR/tools/gemm_vs_vendor_one.py $TILES $SHAPE 3 > $OUT/p$i.log 2>&1
  echo "pass $i ($grp) exit $?" >> $OUT/summary.txt
done
cd $ROOTDIR
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
def short(n):
    if "Cijk" in n: return "vendor " + n[n.find("MT"):n.find("MT") + 26]
    if "gemm_f32_glds" in n: return n[n.find("gemm_f32_glds"):][:70]
    return None
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/p1/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k: dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open("$OUT/pmc_summary.txt", "w") as o:
    for k in agg:
        o.write(f"== {k}   (profiled launch durations us: {[round(d) for d in dur[k]]})\n")
        for c, v in agg[k].items():
            o.write(f"   {c}: n={len(v)} mean={sum(v)/len(v):.6g}\n")
print(open("$OUT/pmc_summary.txt").read())
PY
find $OUT -name "*.csv" -size +2M -delete
