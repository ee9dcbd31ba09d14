#!/usr/bin/env python3
"""Full-size parity of one fine-tuning step (the reference's shape: 3 x (8,1,160000), T = 499, 11 976 rows) in eval-mode
arithmetic: loss and every parameter gradient, merged-branch engine path vs torch autograd on the CPU oracle (~15 s)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict
from oracle import nomad_oracle as O
B, n, margin = int(os.environ.get("B", 8)), 160000, 1.0
sd = seeded_state_dict(3, qk_gain=3.0)
g = torch.Generator().manual_seed(0)
A, P, N = [(0.1 * torch.randn(B, n, generator=g)).clamp(-1, 1) for _ in range(3)]
torch.set_num_threads(min(32, os.cpu_count() or 1))
t0 = time.perf_counter()
ref_loss, ref = O.triplet_step_grads(sd, A, P, N, margin)
t_cpu = time.perf_counter() - t0
eng = Engine({k: v.clone() for k, v in sd.items()}, 0)
eng.train_enable()
w = torch.cat([A, P, N]).cuda()
eng.train_set_branches([0xFFF] * 3)
emb, layers, saved = eng.embed_train(w)
loss, da, dp, dn = eng.triplet_loss(emb[:B].contiguous(), emb[B:2 * B].contiguous(), emb[2 * B:].contiguous(), margin)
eng.train_zero_grad()
eng.train_backward(w, layers, saved, torch.cat([da, dp, dn]))
got = eng.train_unflatten(eng.train_read(1))
top = max(v.abs().max().item() for v in ref.values())
worst = max(((got[k] - v).abs().max().item() / (v.abs().max().item() + 1e-3 * top), k) for k, v in ref.items())
flat_ref = torch.cat([ref[k].reshape(-1) for k in ref]).double()
flat_got = torch.cat([got[k].reshape(-1) for k in ref]).double()
cos = (flat_ref @ flat_got / (flat_ref.norm() * flat_got.norm())).item()
res = {"rows": 3 * B * 499, "loss_gpu": loss.item(), "loss_cpu": ref_loss.item(), "worst_rel_err": worst[0], "worst_tensor": worst[1],
       "cosine": cos, "cpu_step_s": round(t_cpu, 1)}
print(json.dumps(res))
sys.exit(0 if (abs(loss.item() - ref_loss.item()) < 1e-4 and worst[0] < 2e-3 and cos > 0.999999) else 1)
