#!/usr/bin/env python3
"""configs[3] (nomad.forward() + backward to `estimate`, 2 x 32 x 16384 samples) alone: wall time per step next to the sum of the kernel
durations (run it under `rocprofv3 --kernel-trace --stats` for the latter) - is the step launch-bound or kernel-bound?
Usage: python3 tools/c4_profile.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nomad_amd.nomad import Nomad
from nomad_amd.weights import seeded_state_dict
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = torch.Generator().manual_seed(0)
clean_h = (0.1 * torch.randn(32, 1, 16384, generator=g)).clamp(-1, 1)
noise_h = 0.02 * torch.randn(32, 1, 16384, generator=g)
nmd = Nomad(device=0, weights=seeded_state_dict(0), precision=os.environ.get("PREC", "fp32"))
clean = clean_h.to(nmd.DEVICE)
est0 = (clean + noise_h.to(nmd.DEVICE)).clamp(-1, 1)


def fwd_bwd():
    est = est0.clone().requires_grad_(True)
    nmd.forward(est, clean).backward()
    return est.grad


for _ in range(5):
    fwd_bwd()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    fwd_bwd()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t = time.perf_counter() - t0
print("forward+backward: %.3f ms per step wall, host done issuing after %.3f ms per step (steps %d)" % (1e3 * t / steps, 1e3 * t_issue / steps, steps))
