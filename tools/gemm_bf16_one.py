#!/usr/bin/env python3
"""A few launches of one bf16 GEMM instantiation on one shape (target of rocprofv3 --pmc passes).
Usage: gemm_bf16_one.py [shape] [tile]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict
from gemm_sweep import SHAPES
shape = sys.argv[1] if len(sys.argv) > 1 else "c5_qkv"
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 16
eng = Engine(seeded_state_dict(0), 0, diag=True)
g = torch.Generator().manual_seed(0)
M, N, K, has_b, gelu, has_r = SHAPES[shape]
A = torch.randn(M, K, generator=g).cuda().bfloat16()
W = (torch.randn(N, K, generator=g) * K ** -0.5).cuda().bfloat16()
b = torch.randn(N, generator=g).cuda() if has_b else None
R = torch.randn(M, N, generator=g).cuda().bfloat16() if has_r else None
for _ in range(6):
    eng.diag_gemm_bf16(A, W, b, R, gelu=gelu, tile=tile)
torch.cuda.synchronize()
