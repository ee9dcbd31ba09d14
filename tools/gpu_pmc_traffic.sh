#!/bin/bash
# HBM-side traffic of the bench kernels: separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE do not fit one pass; no
# tracing besides --kernel-trace), then tools/pmc_traffic.py <dir> <tag> turns them into profiles/<tag>_pmc_traffic.json.
TAG=${1:-pmc_traffic}
OUT=gpurun_out/$TAG; mkdir -p $OUT
ROOTDIR=$(pwd)
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $ROOTDIR/$OUT/pmc_$c -o p -- python3 $ROOTDIR/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-also --single-stream > $ROOTDIR/$OUT/pmc_$c.log 2>&1
  echo "pmc $c exit $?" | tee -a $ROOTDIR/$OUT/summary.txt
done
cd $ROOTDIR
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if "nomad" in r["Kernel_Name"] and r["Counter_Name"] == c:
                name = r["Kernel_Name"].replace("void nomad::", "").replace("(nomad::GemmParams)", "").split("(")[0]
                agg[name][c].append(float(r["Counter_Value"]))
out = {}
for k, v in agg.items():
    out[k] = {c: {"launches": len(x), "sum_KB": sum(x), "mean_KB_per_launch": sum(x) / len(x)} for c, x in v.items()}
json.dump(out, open("$OUT/pmc_traffic.json", "w"), indent=1)
for k, v in out.items():
    print(k, {c: round(x["mean_KB_per_launch"] / 1024, 1) for c, x in v.items()}, "MB/launch (raw counter)")
PY
python3 tools/pmc_traffic.py $OUT $TAG | head -3
find $OUT -name "*.csv" -size +6M -delete
