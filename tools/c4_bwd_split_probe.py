#!/usr/bin/env python3
"""Round 6 probe: configs[3] with ONLY the differentiated (estimate) branch in H parts - H training forwards of 32 / H clips and H
backwards on H streams (one context each) - next to the 32-clip clean forward on its own stream.  The backward is a single chain of
small kernels with nothing beside it; does a second chain beside it pay?
Usage: python tools/c4_bwd_split_probe.py [--parts 1,2]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict

ap = argparse.ArgumentParser()
ap.add_argument("--parts", default="1,2,1,2")
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
sd = seeded_state_dict(0)
g = torch.Generator().manual_seed(0)
clean = (0.1 * torch.randn(32, 16384, generator=g)).clamp(-1, 1).cuda()
est = (clean + 0.02 * torch.randn(32, 16384, generator=g).cuda()).clamp(-1, 1)
hw = ((torch.rand(256, 768, generator=g) * 2 - 1) / 768 ** 0.5).cuda()
hb = ((torch.rand(256, generator=g) * 2 - 1) / 768 ** 0.5).cuda()
head = (hw, hb)
one = torch.ones((), device="cuda")
ceng = Engine(sd, 0)
cstream = torch.cuda.Stream()
for H in [int(x) for x in a.parts.split(",")]:
    engs = [Engine(sd, 0) for _ in range(H)]
    streams = [torch.cuda.Stream() for _ in range(H)]
    n = 32 // H
    cur = torch.cuda.current_stream()

    def step():
        cstream.wait_stream(cur)
        with torch.cuda.stream(cstream):
            c_emb, c_layers = ceng.embed(clean, head=head, want_layers=True)
        fw = []
        for h in range(H):
            streams[h].wait_stream(cur)
            with torch.cuda.stream(streams[h]):
                fw.append(engs[h].embed_train(est[h * n:(h + 1) * n], head))
        grads = []
        for h in range(H):
            streams[h].wait_stream(cstream)
            with torch.cuda.stream(streams[h]):
                e_emb, e_layers, saved = fw[h]
                cl = c_layers[:, h * n:(h + 1) * n].contiguous() if H > 1 else c_layers
                ce = c_emb[h * n:(h + 1) * n].contiguous() if H > 1 else c_emb
                loss = engs[h].l1_loss(e_layers, cl, e_emb, ce)
                dl, de = engs[h].l1_loss_backward(e_layers, cl, e_emb, ce, one)
                grads.append(engs[h].embed_backward(est[h * n:(h + 1) * n], e_layers, saved, dl, de, head))
        for s in streams:
            cur.wait_stream(s)
        return torch.cat(grads) / H

    for _ in range(4):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        gr = step()
    torch.cuda.synchronize()
    print(json.dumps({"estimate_parts": H, "forward_backward_ms": round(1e3 * (time.perf_counter() - t0) / a.steps, 3), "grad_abs_max": float(gr.abs().max())}), flush=True)
    for e in engs:
        e.close()
