"""Per-parameter gradient error of the HIP fine-tuning step vs autograd on the CPU oracle (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nomad_amd.weights import seeded_state_dict
from nomad_amd.engine import Engine
from oracle import nomad_oracle as O

B, n, margin = 2, 8000, 1.0
sd = seeded_state_dict(3, qk_gain=3.0)
eng = Engine({k: v.clone() for k, v in sd.items()}, 0)
eng.train_enable()
g = torch.Generator().manual_seed(B)
A, P, N = [(0.1 * torch.randn(B, n, generator=g)).clamp(-1, 1) for _ in range(3)]
ref_loss, ref = O.triplet_step_grads(sd, A, P, N, margin)
eng.train_zero_grad()
outs = [eng.embed_train(w.cuda()) for w in (A, P, N)]
loss, da, dp, dn = eng.triplet_loss(outs[0][0], outs[1][0], outs[2][0], margin)
for w, (emb, layers, saved), d in zip((A, P, N), outs, (da, dp, dn)):
    eng.train_backward(w.cuda(), layers, saved, d)
got = eng.train_unflatten(eng.train_read(1))
print("loss", loss.item(), ref_loss.item())
rows = []
for k, want in ref.items():
    s = want.abs().max().item()
    rows.append(((got[k] - want).abs().max().item() / max(s, 1e-30), s, got[k].abs().max().item(), k))
rows.sort(reverse=True)
for e, s, gs, k in rows[:40]:
    print(f"{e:10.3e} ref_max {s:10.3e} got_max {gs:10.3e} {k}")
print("...")
for e, s, gs, k in rows[-5:]:
    print(f"{e:10.3e} ref_max {s:10.3e} got_max {gs:10.3e} {k}")
