#!/bin/bash
# round 3, hunt 5: the shipped build (no packed-FP32 instructions) against the reproducers, and what it costs
mkdir -p gpurun_out
( HUNT_PROVENANCE=0 timeout 200 python tools/race_hunt_conv0.py 8 gemm1,bt5,bt11,bt12,forward ) > gpurun_out/h5_micro_shipped.log 2>&1
( NOMAD_LIB_VARIANT=pk HUNT_PROVENANCE=0 timeout 100 python tools/race_hunt_conv0.py 5 bt11,bt12 ) > gpurun_out/h5_micro_pk.log 2>&1
grep "^variant" gpurun_out/h5_micro_shipped.log gpurun_out/h5_micro_pk.log
( NOMAD_LIB_VARIANT=pk timeout 400 python tools/race_hunt_forward.py 7 bf16,fp32,bf16x3 none,bt1,bt5,bt11,bt12,forward16 ) > gpurun_out/h5_forward_pk.log 2>&1
( timeout 400 python tools/race_hunt_forward.py 7 bf16,fp32,bf16x3 none,bt1,bt5,bt11,bt12,forward16 ) > gpurun_out/h5_forward_shipped.log 2>&1
grep "^library\|first stage" gpurun_out/h5_forward_pk.log gpurun_out/h5_forward_shipped.log
( timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > gpurun_out/h5_bench_shipped.json 2> gpurun_out/h5_bench_shipped.err
( NOMAD_LIB_VARIANT=pk timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > gpurun_out/h5_bench_pk.json 2> gpurun_out/h5_bench_pk.err
python - <<'PY'
import json
for n in ("shipped", "pk"):
    try:
        d = json.load(open(f"gpurun_out/h5_bench_{n}.json"))
        print(n, "value", d["value"], "roofline", d["roofline"]["frac"], "x3", d.get("also_measured", {}).get("value"), "c5", d.get("also_measured_c5", {}).get("value"),
              "c3", d.get("also_measured_c3"), "c4", d.get("also_measured_c4"), "peaky", d.get("also_measured_peaky"), "ktimes", d.get("kernel_time_ms_per_step"))
    except Exception as e:
        print(n, "bench failed", e); print(open(f"gpurun_out/h5_bench_{n}.err").read()[-1500:])
PY
