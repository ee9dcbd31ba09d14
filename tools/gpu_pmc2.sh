#!/bin/bash
TAG=${1:-pmc2}
bash tools/gpu_pmc.sh $TAG/t21_fc1 21 fc1_nogelu > /dev/null 2>&1
bash tools/gpu_pmc.sh $TAG/t29_fc2 29 fc2 > /dev/null 2>&1
bash tools/gpu_pmc.sh $TAG/t21_conv3 21 conv3 > /dev/null 2>&1
for d in t21_fc1 t29_fc2 t21_conv3; do echo "== $d"; cat gpurun_out/$TAG/$d/pmc_summary.txt; done
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline
