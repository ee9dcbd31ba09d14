#!/bin/bash
TAG=${1:-pmc3}
bash tools/gpu_pmc.sh $TAG/t33_fc1 33 fc1 > /dev/null 2>&1
bash tools/gpu_pmc.sh $TAG/t33_conv3 33 conv3 > /dev/null 2>&1
bash tools/gpu_pmc.sh $TAG/t34_out 34 out > /dev/null 2>&1
for d in t33_fc1 t33_conv3 t34_out; do
python3 - <<PY
import csv, re
d="gpurun_out/$TAG/$d"
rows=[r for r in csv.DictReader(open(d+"/p4/p_kernel_trace.csv")) if "gemm" in r["Kernel_Name"]]
durs=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows]
txt=open(d+"/pmc_summary.txt").read()
def val(name):
    m=re.search(name+r": n=\d+ mean=([0-9.e+]+)", txt); return float(m.group(1)) if m else float("nan")
gui=val("GRBM_GUI_ACTIVE"); busy=val("SQ_VALU_MFMA_BUSY_CYCLES"); dur=sum(durs)/len(durs)
line=f"$d: dur_us={dur:.0f} (profiled) clock_GHz={gui/8/dur/1e3:.3f} mfma_busy_frac={busy/1024/(gui/8):.3f} lds_bank_conflict={val('SQ_LDS_BANK_CONFLICT'):.0f} fetch_KB={val('FETCH_SIZE'):.0f} write_KB={val('WRITE_SIZE'):.0f} tcc_hit_rate={val('TCC_HIT_sum')/(val('TCC_HIT_sum')+val('TCC_MISS_sum')):.3f}"
print(line); open(d+"/derived.txt","w").write(line+"\n")
PY
done
