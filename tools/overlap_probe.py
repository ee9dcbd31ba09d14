#!/usr/bin/env python3
"""Experiment: does running two half-batches on two HIP streams beat one full batch? (tail/gap overlap)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict

sd = seeded_state_dict(0)
e = [Engine(sd, 0), Engine(sd, 0)]
g = torch.Generator().manual_seed(0)
wav = (0.1 * torch.randn(256, 64000, generator=g)).clamp(-1, 1).cuda()
st = [torch.cuda.Stream(), torch.cuda.Stream()]

MODE = sys.argv[1] if len(sys.argv) > 1 else "fp32"   # fp32 | bf16x3


def emb(eng, w):
    return eng.embed_bf16x3(w) if MODE == "bf16x3" else eng.embed(w)


def single():
    return emb(e[0], wav)

def dual(parts=2):
    outs = []
    cur = torch.cuda.current_stream()
    chunk = 256 // parts
    for i in range(parts):
        s = st[i % 2]
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(emb(e[i % 2], wav[i * chunk:(i + 1) * chunk]))
    for s in st:
        cur.wait_stream(s)
    return torch.cat(outs)

ref = single().clone()
for name, fn in (("single B=256", single), ("2 streams x 128", dual), ("2 streams, 4 x 64", lambda: dual(4)), ("single B=256", single),
                 ("2 streams x 128", dual), ("2 streams, 4 x 64", lambda: dual(4))):
    for _ in range(2):
        out = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{name:18s} {1e3 * dt:8.2f} ms/step  {256 / dt:8.1f} clips/s  bit-identical={bool(torch.equal(out, ref))}")
