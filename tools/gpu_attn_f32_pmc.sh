#!/bin/bash
# PMC passes over the bf16 attention micro harness (all variants it runs).  Usage: bash tools/gpu_attn_pmc.sh <tag>
TAG=${1:-attnpmc}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -o /tmp/attn_f32 tools/micro/attn_f32.hip > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VALU_TRANS" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o p -- /tmp/attn_f32 256 199 1 > $OUT/p$i.log 2>&1
  echo "pass $i ($grp) exit $?" >> $OUT/summary.txt
done
cd $ROOTDIR
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attention_f32" in k:
            k = k.replace("void nomad::", "").split("(")[0]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/p1/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attention_f32" in k:
            dur[k.replace("void nomad::", "").split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open("$OUT/pmc_summary.txt", "w") as o:
    for k, d in agg.items():
        o.write(k + "  dur_us(mean, profiled) %.1f\n" % (sum(dur[k]) / max(1, len(dur[k]))))
        for c, v in d.items():
            o.write("   %-28s %.6g\n" % (c, sum(v) / len(v)))
print(open("$OUT/pmc_summary.txt").read())
PY
find $OUT -name "*.csv" -size +2M -delete
