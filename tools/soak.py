#!/usr/bin/env python3
"""Long race screen: full 256 x 4 s batch, fp32 / bf16 / ragged, repeated; every run must be bit-identical."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict
eng = Engine(seeded_state_dict(0), 0)
g = torch.Generator().manual_seed(0)
wav = (0.1 * torch.randn(256, 64000, generator=g)).clamp(-1, 1).cuda()
lens = torch.randint(8000, 100000, (96,), generator=g).tolist()
rag = [(0.1 * torch.randn(n, generator=g)).clamp(-1, 1).cuda() for n in lens]
r32, r16, rr = eng.embed(wav).clone(), eng.embed_bf16(wav).clone(), eng.embed_ragged(rag).clone()
bad = 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for i in range(n):
    bad += int(not torch.equal(eng.embed(wav), r32))
    if i % 3 == 0:
        bad += int(not torch.equal(eng.embed_bf16(wav), r16))
        bad += int(not torch.equal(eng.embed_ragged(rag), rr))
torch.cuda.synchronize()
print(f"soak: {n} iterations, mismatches = {bad}")
sys.exit(1 if bad else 0)
