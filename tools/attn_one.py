#!/usr/bin/env python3
"""Run and time the fp32 attention kernel on the bench shape (256 clips x 12 heads x T=199; B / T / SCALE from the environment) -
also the target for --pmc passes.  DIAG=1 loads libnomad_diag.so, where NOMAD_ATTN_PIPE=1 selects the experimental persistent kernel
(attention_f32_v3.hip.h) and NOMAD_ATTN_ABLATE=<bits> its timing probes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict
B, T = int(os.environ.get("B", 256)), int(os.environ.get("T", 199))
eng = Engine(seeded_state_dict(0), 0, diag=bool(int(os.environ.get("DIAG", "0"))))
qkv = (torch.randn(B * T, 2304, generator=torch.Generator().manual_seed(0)) * float(os.environ.get("SCALE", "0.5"))).cuda()
for _ in range(3):
    eng.diag_attention(qkv, B, T)
torch.cuda.synchronize()
n = int(os.environ.get("ITERS", 20))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for i in range(n):
    eng.diag_attention(qkv, B, T)
    ev[i + 1].record()
torch.cuda.synchronize()
ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
fl = 4.0 * B * 12 * T * T * 64
print(os.environ.get("NOMAD_ATTN_ABLATE", "-"), os.environ.get("NOMAD_ATTN_PIPE", "-"), "attention fp32 B=%d T=%d: median %.4f ms, min %.4f ms, %.1f TFLOP/s (useful flops 4*B*12*T*T*64)" % (B, T, ts[n // 2], ts[0], fl / ts[n // 2] / 1e9))
