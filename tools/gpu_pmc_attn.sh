#!/bin/bash
# PMC passes for the fp32 attention kernel.  Usage: bash tools/gpu_pmc_attn.sh <tag>
TAG=${1:-pmca}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o p -- python3 $ROOTDIR/tools/attn_one.py > $OUT/p$i.log 2>&1
  echo "pass $i ($grp) exit $?" >> $OUT/summary.txt
done
cd $ROOTDIR
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "attention_f32" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/pmc_summary.txt", "w") as o:
    for k, v in agg.items():
        line = f"{k}: n={len(v)} mean={sum(v)/len(v):.6g}"
        print(line); o.write(line + "\n")
for f in sorted(glob.glob("$OUT/p1/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "attention_f32" in r["Kernel_Name"]:
            print("dur_us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
PY
find $OUT -name "*.csv" -size +4M -delete
