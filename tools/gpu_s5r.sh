#!/bin/bash
# round 5, trip r: PMC passes of the bf16 attention kernels at configs[4]'s shape (v3 = 16x16x32 shipped, v2 = 32x32x16): instruction mix and
# busy / wait cycles - the evidence for "bound by vector-instruction issue" (separate rocprofv3 --pmc passes, kernel-trace only)
TAG=${1:-s5r}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
GRPS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY"
      "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"
      "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
      "SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
      "GRBM_GUI_ACTIVE SQ_WAVES")
for v in 2 0; do
  export NOMAD_BF16_ATTN_V3=$v
  i=0
  for grp in "${GRPS[@]}"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/v$v/p$i -o p -- python3 $ROOTDIR/tools/attn_bf16_ab.py --iters 6 > $OUT/v${v}_p$i.log 2>&1
    echo "v$v pass $i ($grp) exit $?" >> $OUT/summary.txt
  done
done
cd $ROOTDIR
python3 - <<PY
import csv, glob, collections
with open("$OUT/pmc_summary.txt", "w") as o:
    for v in (2, 0):
        agg = collections.defaultdict(list); dur = []
        for f in sorted(glob.glob("$OUT/v%d/p*/**/*counter_collection.csv" % v, recursive=True)):
            for r in csv.DictReader(open(f)):
                if "attention_bf16" in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for f in sorted(glob.glob("$OUT/v%d/p1/**/*kernel_trace.csv" % v, recursive=True)):
            for r in csv.DictReader(open(f)):
                if "attention_bf16" in r["Kernel_Name"]:
                    dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        line = "NOMAD_BF16_ATTN_V3=%d, per launch (B = 32, T = 1499); duration under the profiler (us): median %.1f" % (v, sorted(dur)[len(dur) // 2] if dur else -1)
        print(line); o.write(line + "\n")
        for k, vals in agg.items():
            line = f"  {k}: n={len(vals)} mean={sum(vals)/len(vals):.6g}"
            print(line); o.write(line + "\n")
PY
cat $OUT/summary.txt | tail -3
rm -rf $OUT/v2 $OUT/v0
