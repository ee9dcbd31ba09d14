#!/usr/bin/env python3
"""The fp32 attention kernel alone at the headline's shape (256 clips x T = 199; --B / --T for others): time per launch, TFLOP/s, max
error of clip 0 against a float64 softmax(QK^T)V, and a checksum of the whole output (equal across builds = bit-identical)."""
import argparse, hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=256)
    ap.add_argument("--T", type=int, default=199)
    ap.add_argument("--iters", type=int, default=40)
    a = ap.parse_args()
    eng = Engine(seeded_state_dict(0), 0, diag=True)
    g = torch.Generator().manual_seed(1)
    B, T = a.B, a.T
    qkv = torch.randn(B * T, 2304, generator=g)
    qkv[:, :1536] *= 0.35
    x = qkv.cuda()
    out = eng.diag_attention(x, B, T)
    torch.cuda.synchronize()
    xd = x[:T].double().cpu()
    q, k, v = (xd[:, i * 768:(i + 1) * 768].view(1, T, 12, 64).transpose(1, 2) for i in range(3))
    ref = (torch.softmax(q @ k.transpose(-1, -2), -1) @ v).transpose(1, 2).reshape(T, 768)
    err = float((out[:T].double().cpu() - ref).abs().max())
    digest = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.iters + 1)]
    ev[0].record()
    for i in range(a.iters):
        eng.diag_attention(x, B, T)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(a.iters))
    med = ms[len(ms) // 2]
    fl = 4.0 * B * 12 * T * T * 64
    print(json.dumps({"B": B, "T": T, "us_median": round(med * 1e3, 1), "us_min": round(ms[0] * 1e3, 1), "tflops": round(fl / med / 1e9, 1),
                      "max_abs_err_clip0": err, "sha256_16": digest}), flush=True)


if __name__ == "__main__":
    main()
