#!/usr/bin/env python3
"""Run the fp32 attention kernel on the bench shape (256 clips x 12 heads x T=199) a few times (target for --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict
B, T = int(os.environ.get("B", 256)), int(os.environ.get("T", 199))
eng = Engine(seeded_state_dict(0), 0)
qkv = (torch.randn(B * T, 2304, generator=torch.Generator().manual_seed(0)) * 0.5).cuda()
for _ in range(3):
    eng.diag_attention(qkv, B, T)
torch.cuda.synchronize()
print("done")
