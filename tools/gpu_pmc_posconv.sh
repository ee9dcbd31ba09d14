#!/bin/bash
# PMC passes of the slab pos-conv at configs[4]'s shape (one rocprofv3 run per counter group, kernel-trace only)
TAG=${1:-pmcposconv}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
GRPS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY"
      "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"
      "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
      "GRBM_GUI_ACTIVE GRBM_COUNT"
      "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"
      "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"
      "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"
      "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum")
for grp in "${GRPS[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o p -- python3 $ROOTDIR/tools/posconv_one.py 1 > $OUT/p$i.log 2>&1
  echo "pass $i ($grp) exit $?" >> $OUT/summary.txt
done
cd $ROOTDIR
python3 - <<PY > $OUT/pmc_posconv.txt
import csv, glob, collections
agg = collections.defaultdict(list); dur = []
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "posconv_bf16_slab" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in sorted(glob.glob("$OUT/p1/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "posconv_bf16_slab" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("posconv_bf16_slab_kernel<8, 1>, 32 x T = 1499: launches", len(dur), "durations us", [round(d, 1) for d in dur])
for k in sorted(agg):
    v = agg[k]
    print("%-34s per launch %.4g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
cat $OUT/pmc_posconv.txt
rm -rf $OUT/p[0-9]*
