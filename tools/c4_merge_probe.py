#!/usr/bin/env python3
"""Round 6 probe: the two forward branches of configs[3] (estimate, clean: 32 clips of 16384 samples each) as two concurrent 32-clip
forwards on two streams (what Nomad.forward does) against ONE 64-clip forward (same weights, twice the rows per GEMM launch, half the
launches).  Layer outputs wanted in both cases.
Usage: python tools/c4_merge_probe.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict

eng = Engine(seeded_state_dict(0), 0)
g = torch.Generator().manual_seed(0)
wav = (0.1 * torch.randn(64, 16384, generator=g)).clamp(-1, 1).cuda()
a, b = wav[:32].contiguous(), wav[32:].contiguous()
side = eng.side_stream()
cur = torch.cuda.current_stream()


def two():
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        r1 = eng.embed(b, want_layers=True, side=True)
    r0 = eng.embed(a, want_layers=True)
    cur.wait_stream(side)
    return r0, r1


def one():
    return eng.embed(wav, want_layers=True)


def timeit(fn, n=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


for rep in range(2):
    print(json.dumps({"two_concurrent_32_clip_forwards_ms": round(timeit(two), 3), "one_64_clip_forward_ms": round(timeit(one), 3)}), flush=True)
(e0, l0), (e1, l1) = two()
e, l = one()
torch.cuda.synchronize()
print(json.dumps({"embeddings_bit_equal": bool(torch.equal(torch.cat([e0, e1]), e)), "layers_max_abs_diff": float((torch.cat([l0, l1], 1) - l).abs().max())}))
