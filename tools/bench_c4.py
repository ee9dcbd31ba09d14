#!/usr/bin/env python3
"""Config C4 (BASELINE.json configs[3]): nomad.forward() as an auxiliary loss inside a training step -
per-step latency of loss forward and forward+backward at the reference example's shapes
(nomad_loss_test.py: batch 32, clips zero-padded/cropped to 16384 samples, T = 50)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from nomad_amd.nomad import Nomad  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--samples", type=int, default=16384)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", default="fp32", help="bf16x3: the no-gradient branch(es) run the split-bf16 forward")
    a = ap.parse_args()
    nmd = Nomad(weights="seeded", precision=a.precision)
    g = torch.Generator().manual_seed(0)
    clean = (0.1 * torch.randn(a.batch, 1, a.samples, generator=g)).clamp(-1, 1).cuda()
    est0 = (clean + 0.02 * torch.randn(a.batch, 1, a.samples, generator=g).cuda()).clamp(-1, 1)

    def fwd():
        return nmd.forward(est0, clean)

    def fwd_bwd():
        est = est0.clone().requires_grad_(True)
        loss = nmd.forward(est, clean)
        loss.backward()
        return est.grad

    out = {"config": f"C4: nomad.forward() on 2x({a.batch},1,{a.samples}), {a.precision}, 1 GPU", "steps": a.steps}
    for name, fn in (("forward_ms", fwd), ("forward_backward_ms", fwd_bwd)):
        for _ in range(a.warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            r = fn()
        torch.cuda.synchronize()
        out[name] = round(1e3 * (time.perf_counter() - t0) / a.steps, 3)
        assert torch.isfinite(r).all()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
