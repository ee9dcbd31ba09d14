#!/usr/bin/env python3
"""bf16x3 attention: the tiled kernel against the K/V-resident one (4 / 8 waves per (clip, head)) on the bench shape."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict
eng = Engine(seeded_state_dict(0), 0)
g = torch.Generator().manual_seed(0)
for B, T in ((256, 199), (256, 49), (64, 249)):
    qkv = torch.randn(B * T, 2304, generator=g)
    qkv[:, :1536] *= 0.35
    qs = eng.diag_split_bf16(qkv.cuda())
    ref = None
    for waves in (0, 4, 8, 0, 4, 8):
        out = eng.diag_attention_bf16x3(qs, B, T, waves=waves)
        ref = out if ref is None else ref
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
        ev[0].record()
        for i in range(20):
            eng.diag_attention_bf16x3(qs, B, T, waves=waves)
            ev[i + 1].record()
        torch.cuda.synchronize()
        ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(20))
        print(json.dumps({"B": B, "T": T, "waves": waves, "us_med": round(1e3 * ms[10], 1), "bit_identical": bool(torch.equal(out, ref))}), flush=True)
