#!/usr/bin/env python3
"""A few launches of one bf16x3 GEMM shape (for rocprofv3 --pmc passes).  Usage: gemm_x3_one.py [shape] [variant]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict
from gemm_sweep import SHAPES
shape = sys.argv[1] if len(sys.argv) > 1 else "qkv"
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 8
eng = Engine(seeded_state_dict(0), 0, diag=True)  # libnomad_diag.so: experimental tile ids
g = torch.Generator().manual_seed(0)
M, N, K, has_b, gelu, has_r = SHAPES[shape]
A = eng.diag_split_bf16(torch.randn(M, K, generator=g).cuda())
W = eng.diag_split_bf16((torch.randn(N, K, generator=g) * K ** -0.5).cuda())
out = eng.diag_gemm_bf16x3(A, W, variant=variant)
for _ in range(4):
    eng.diag_gemm_bf16x3(A, W, variant=variant, out=out)
torch.cuda.synchronize()
