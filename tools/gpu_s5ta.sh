#!/bin/bash
# round 5: texture-address unit busy time of the persistent bf16 GEMM (tile 60) on configs[4]'s shapes (a few PMC groups only)
TAG=${1:-s5ta}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for SHAPE in c5_qkv c5_fc1 c5_fc2 c5_out c5_conv4; do
  i=0
  for grp in "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/${SHAPE}_p$i -o p -- python3 $ROOTDIR/tools/gemm_bf16_one.py $SHAPE 60 > $OUT/${SHAPE}_p$i.log 2>&1
  done
done
cd $ROOTDIR
python3 - <<PY > $OUT/ta_summary.txt
import csv, glob, collections
for shape in ("c5_qkv", "c5_fc1", "c5_fc2", "c5_out", "c5_conv4"):
    agg = collections.defaultdict(list); dur = []
    for f in sorted(glob.glob("$OUT/%s_p*/**/*counter_collection.csv" % shape, recursive=True)):
        for r in csv.DictReader(open(f)):
            if "gemm_bf16_p9" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in sorted(glob.glob("$OUT/%s_p1/**/*kernel_trace.csv" % shape, recursive=True)):
        for r in csv.DictReader(open(f)):
            if "gemm_bf16_p9" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    d = sum(dur) / max(1, len(dur))
    print(shape, "duration us (under the profiler) %.1f" % d, {k: "%.4g" % (sum(v) / len(v)) for k, v in sorted(agg.items())})
PY
cat $OUT/ta_summary.txt
find $OUT -name "*.csv" -delete
