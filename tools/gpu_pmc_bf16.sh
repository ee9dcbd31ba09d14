#!/bin/bash
# PMC passes for the bf16 8-phase GEMM on a config-C5 shape (one rocprofv3 run per counter group, kernel-trace only).
# Usage: [PMC_SET=cache] bash tools/gpu_pmc_bf16.sh <tag> [shape] [tile]      (PMC_SET=cache: only the L2 / fabric traffic groups)
TAG=${1:-pmcbf16}; SHAPE=${2:-c5_qkv}; TILE=${3:-16}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
GROUPS_ALL=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY"
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
           "GRBM_GUI_ACTIVE GRBM_COUNT"
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum"
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum"
           "FETCH_SIZE" "WRITE_SIZE")
GROUPS_CACHE=("TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"
              "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
              "FETCH_SIZE" "WRITE_SIZE")
if [ "$PMC_SET" = cache ]; then GRPS=("${GROUPS_CACHE[@]}"); else GRPS=("${GROUPS_ALL[@]}"); fi
for grp in "${GRPS[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o p -- python3 $ROOTDIR/tools/gemm_bf16_one.py $SHAPE $TILE > $OUT/p$i.log 2>&1
  echo "pass $i ($grp) exit $?" >> $OUT/summary.txt
done
cd $ROOTDIR
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
dur = []
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "gemm_bf16" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in sorted(glob.glob("$OUT/p1/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "gemm_bf16" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open("$OUT/pmc_summary.txt", "w") as o:
    o.write("shape $SHAPE tile $TILE (bf16 GEMM), per launch; kernel duration under the profiler (us): %s\n" % ", ".join("%.1f" % d for d in dur))
    for k, v in agg.items():
        line = f"{k}: n={len(v)} mean={sum(v)/len(v):.6g}"
        print(line); o.write(line + "\n")
PY
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +4M -delete
