#!/usr/bin/env python3
"""Shader clock held during the config-C5 bf16 forward (32 x 30 s), by the one-wave probe of tools/clock_under_load.py on a side
stream, with the time per forward next to it.  Run once per NOMAD_BF16_N192 setting (the switch is read once per process).
Usage: python3 tools/clock_c5.py [split|single]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict
eng = Engine(seeded_state_dict(0), 0, diag=True)
if len(sys.argv) > 1 and sys.argv[1] == "single":
    eng.BF16_SPLIT_ROWS = 0
side = torch.cuda.Stream()
g = torch.Generator().manual_seed(0)
wav = (0.1 * torch.randn(32, 480000, generator=g)).clamp(-1, 1).cuda()
for _ in range(3):
    eng.embed_bf16(wav)
torch.cuda.synchronize()
res = {"mode": sys.argv[1] if len(sys.argv) > 1 else "split", "n192": os.environ.get("NOMAD_BF16_N192", "auto")}
for rep in range(3):
    n = 60
    for _ in range(5):
        eng.embed_bf16(wav)                   # load in flight
    out = eng.diag_clock_probe(600, side)     # 600 ms of wall clock on the side stream
    t0 = time.time()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        eng.embed_bf16(wav)
    e1.record()
    torch.cuda.synchronize()
    cyc, ticks = out.tolist()
    res[f"rep{rep}"] = {"shader_mhz": round(cyc / (ticks / 100.0), 1), "ms_per_forward": round(e0.elapsed_time(e1) / n, 3)}
print(json.dumps(res))
