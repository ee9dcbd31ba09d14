#!/usr/bin/env python3
"""Round 6 probe for configs[3] (nomad.forward() + backward, 2 x (32,1,16384)): does the step get faster when the batch is run as
H independent parts (32 / H clips each) on H streams - every part its own forward pair (estimate: training forward, clean: plain
forward with layer outputs), loss and backward?  The parts use H separate contexts here (a context's training scratch is not
shared between concurrent calls); results are compared with the one-part step (loss = mean of the parts' losses, gradient =
concatenation / H).
Usage: python tools/c4_split_probe.py [--parts 1,2,4] [--steps 20]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from nomad_amd.engine import Engine  # noqa: E402
from nomad_amd.weights import seeded_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--parts", default="1,2,4")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--samples", type=int, default=16384)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    a = ap.parse_args()
    sd = seeded_state_dict(0)
    g = torch.Generator().manual_seed(0)
    clean = (0.1 * torch.randn(a.batch, a.samples, generator=g)).clamp(-1, 1).cuda()
    est = (clean + 0.02 * torch.randn(a.batch, a.samples, generator=g).cuda()).clamp(-1, 1)
    hw = ((torch.rand(256, 768, generator=g) * 2 - 1) / 768 ** 0.5).cuda()
    hb = ((torch.rand(256, generator=g) * 2 - 1) / 768 ** 0.5).cuda()
    head = (hw, hb)
    one = torch.ones((), device="cuda")
    ref = None
    for H in [int(x) for x in a.parts.split(",")]:
        engs = [Engine(sd, 0) for _ in range(H)]
        streams = [torch.cuda.Stream() for _ in range(H)]
        n = a.batch // H
        cur = torch.cuda.current_stream()

        def step():
            losses, grads = [], []
            for h in range(H):
                e = engs[h]
                s = streams[h]
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    side = e.side_stream()
                    side.wait_stream(s)
                    with torch.cuda.stream(side):
                        c_emb, c_layers = e.embed(clean[h * n:(h + 1) * n], head=head, want_layers=True, side=True)
                    e_emb, e_layers, saved = e.embed_train(est[h * n:(h + 1) * n], head)
                    s.wait_stream(side)
                    loss = e.l1_loss(e_layers, c_layers, e_emb, c_emb)
                    dl, de = e.l1_loss_backward(e_layers, c_layers, e_emb, c_emb, one)
                    dw = e.embed_backward(est[h * n:(h + 1) * n], e_layers, saved, dl, de, head)
                    losses.append(loss)
                    grads.append(dw)
            for s in streams:
                cur.wait_stream(s)
            return torch.stack([l.reshape(()) for l in losses]).mean(), torch.cat(grads) / H

        for _ in range(a.warmup):
            loss, grad = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            loss, grad = step()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / a.steps
        row = {"parts": H, "clips_per_part": n, "forward_backward_ms": round(ms, 3), "loss": float(loss)}
        if ref is None:
            ref = (float(loss), grad.clone())
        else:
            row["loss_rel_diff_vs_1_part"] = abs(float(loss) - ref[0]) / abs(ref[0])
            row["grad_rel_diff_vs_1_part"] = float((grad - ref[1]).abs().max() / ref[1].abs().max())
        print(json.dumps(row), flush=True)
        for e in engs:
            e.close()


if __name__ == "__main__":
    main()
