#!/bin/bash
TAG=${1:-r01g}
OUT=gpurun_out/$TAG; mkdir -p $OUT
ROOTDIR=$(pwd)
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -n 5 $OUT/pytest_gpu.log
timeout 600 python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" | tee -a $OUT/summary.txt
cat $OUT/bench.json
# the N>1 launch path at world size 1: RCCL init + all-gather on the real backend
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_torchrun.json 2> $OUT/bench_torchrun.err; echo "torchrun bench exit $?" | tee -a $OUT/summary.txt
cat $OUT/bench_torchrun.json; tail -n 3 $OUT/bench_torchrun.err
# HBM traffic of the bench kernels: separate PMC passes (FETCH_SIZE / WRITE_SIZE do not fit one pass)
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $ROOTDIR/$OUT/pmc_$c -o p -- python3 $ROOTDIR/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-also > $ROOTDIR/$OUT/pmc_$c.log 2>&1
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
find $OUT -name "*.csv" -size +6M -delete
