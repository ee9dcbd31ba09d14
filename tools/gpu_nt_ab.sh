#!/bin/bash
# The bf16 8-phase GEMM's epilogue on config C5's shapes: non-temporal output stores / residual loads (diag tiles 42 / 43 / 44)
# against plain ones (tile 16 with NOMAD_BF16_NT_STORES=0), what the GELU costs (the *_nogelu shapes; tile 17 = no stores), and
# the L2 / fabric counters of both store flavours.  -> profiles/r03_pmc_gemm_bf16_nt_stores.txt
mkdir -p gpurun_out/nt_ab
export NOMAD_BF16_NT_STORES=0     # tile 16 = the plain-store kernel in this script
python3 tools/gemm_sweep.py --bf16 --tiles 16,16,42,43,44,16,42,43,44 --shapes c5_out,c5_fc2,c5_qkv,c5_fc1,c5_conv4,c5_conv2 --iters 20 --json gpurun_out/nt_ab/sweep2.json > gpurun_out/nt_ab/sweep2.log 2>&1
python3 tools/gemm_sweep.py --bf16 --tiles 16,16,17,16,17 --shapes c5_fc1,c5_fc1_nogelu,c5_conv4,c5_conv4_nogelu,c5_qkv --iters 20 --json gpurun_out/nt_ab/sweep3.json > gpurun_out/nt_ab/sweep3.log 2>&1
for f in sweep2 sweep3; do grep -o '"shape": "[a-z0-9_]*", "tile": [0-9]*\|"ms_med": [0-9.]*\|"tflops": [0-9.]*' gpurun_out/nt_ab/$f.log | paste - - - ; done
for sh in c5_qkv c5_out; do for t in 16 42; do
  PMC_SET=cache bash tools/gpu_pmc_bf16.sh nt_ab/pmc_${sh}_t$t $sh $t > /dev/null 2>&1
  echo "== $sh tile $t"; cat gpurun_out/nt_ab/pmc_${sh}_t$t/pmc_summary.txt
done; done
