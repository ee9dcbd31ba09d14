#!/usr/bin/env python3
"""Background load for hazard screening: loops the kernels that exposed the packed-FP32 hazard (bf16 128 x 128 GEMM variants,
diag tiles 1 / 5 / 11 / 12) plus a bf16 forward and rocBLAS matmuls on two streams until killed or for `seconds`.
Usage: python tools/aggressor.py [seconds]   (run it next to `pytest -m gpu`: every parity test must still pass)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1e9
eng = Engine(seeded_state_dict(0), 0, diag=True)
g = torch.Generator().manual_seed(1)
A = (torch.randn(6368, 768, generator=g) * 0.5).to(torch.bfloat16).cuda()
W = (torch.randn(768, 768, generator=g) * 0.03).to(torch.bfloat16).cuda()
wav = (0.1 * torch.randn(32, 64000, generator=g)).clamp(-1, 1).cuda()
junk = torch.randn(2048, 2048, device="cuda")
s2 = torch.cuda.Stream()
t_end = time.time() + secs
n = 0
while time.time() < t_end:
    for t in (11, 12, 1, 5):
        for _ in range(4):
            eng.diag_gemm_bf16(A, W, tile=t)
    with torch.cuda.stream(s2):
        eng.embed_bf16(wav)
        junk = junk @ junk * 1e-3
    n += 1
    if n % 20 == 0:
        torch.cuda.synchronize()
print("aggressor loops", n)
