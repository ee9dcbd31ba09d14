#!/usr/bin/env python3
"""Does stream capture (hipGraph via torch.cuda.CUDAGraph) work around the engine's launches, and what does it buy at
small batch where kernels are tens of microseconds?"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict
eng = Engine(seeded_state_dict(0), 0)
g = torch.Generator().manual_seed(0)
res = {}
for B, n in ((1, 64000), (4, 64000), (32, 16384)):
    wav = (0.1 * torch.randn(B, n, generator=g)).clamp(-1, 1).cuda()
    ref = eng.embed(wav).clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        out = eng.embed(wav)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 50 * 1e3
    static_in = wav.clone()
    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        eng.embed(static_in)  # warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(graph):
        static_out = eng.embed(static_in)
    static_in.copy_(wav)
    graph.replay()
    torch.cuda.synchronize()
    same = torch.equal(static_out, ref)
    t0 = time.perf_counter()
    for _ in range(50):
        graph.replay()
    torch.cuda.synchronize()
    gr = (time.perf_counter() - t0) / 50 * 1e3
    res[f"B{B}_n{n}"] = {"eager_ms": round(eager, 3), "graph_ms": round(gr, 3), "bit_identical": same}
    print(f"B={B} n={n}: eager {eager:.3f} ms, graph replay {gr:.3f} ms, identical {same}", flush=True)
print(json.dumps(res))
