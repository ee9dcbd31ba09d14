#!/bin/bash
# round 4: hazard reproducers + the bench line with its new legs
TAG=${1:-r4p}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -o /tmp/store_hazard tools/micro/store_hazard.hip 2>/dev/null && timeout 300 /tmp/store_hazard > $OUT/store_hazard.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_fma_hazard tools/micro/pk_fma_hazard.hip 2>/dev/null && timeout 300 /tmp/pk_fma_hazard > $OUT/pk_fma_hazard.txt 2>&1
cat $OUT/store_hazard.txt $OUT/pk_fma_hazard.txt
timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
python3 - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"]["traffic"])
print("cpu", d.get("cpu_baseline"))
print("c3", {k: v for k, v in (d.get("also_measured_c3") or {}).items() if k != "workload"})
print("c5", {k: v for k, v in (d.get("also_measured_c5") or {}).items() if k not in ("workload", "context")})
print("c4", {k: v for k, v in (d.get("also_measured_c4") or {}).items() if k != "workload"})
PY
tail -3 $OUT/bench.err
