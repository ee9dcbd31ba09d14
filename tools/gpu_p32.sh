#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "gemm_bf16" 2>&1 | tail -3
python tools/gemm_sweep.py --bf16 --tiles 16,40,17,41 --shapes c5_out,c5_qkv,c5_fc1,c5_fc2,c5_conv4,sq4096 --iters 7 --json gpurun_out/r3_gemm_sweep_bf16_mfma32.json 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['shape'], r['tile'], r['cfg'], r['ms_med'], r['tflops'], 'rel_diff_vs_16', r['rel_diff_vs_first'])"
for v in 1 0 1 0; do NOMAD_BF16_8PHASE_MFMA32=$v python bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('mfma32=$v', d['value'], 'clips/s', d['kernel_time_ms_per_step'])"; done
