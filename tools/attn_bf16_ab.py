#!/usr/bin/env python3
"""Round 5: the bf16 attention kernel alone at config C5's shape (32 clips x T = 1499) - time per launch, TFLOP/s, max error against a
float64 softmax(QK^T)V on the bf16 inputs.  The kernel is chosen by NOMAD_BF16_ATTN_V3 (diag library): 1 = 16x16x32 (round 5), 0 = 32x32x16.
Usage: NOMAD_BF16_ATTN_V3=0|1 python tools/attn_bf16_ab.py [--B 32 --T 1499]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--T", type=int, default=1499)
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    eng = Engine(seeded_state_dict(0), 0, diag=True)
    g = torch.Generator().manual_seed(1)
    B, T = a.B, a.T
    qkv = torch.randn(B * T, 2304, generator=g)
    qkv[:, :1536] *= 0.35
    qkv[:, :768] *= 1.4426950408889634
    x = qkv.bfloat16().cuda()
    out = eng.diag_attention_bf16(x, B, T, q_has_log2e=True)
    torch.cuda.synchronize()
    # reference on the first clip only (float64)
    xd = x[:T].double().cpu()
    xd[:, :768] /= 1.4426950408889634
    q, k, v = (xd[:, i * 768:(i + 1) * 768].view(1, T, 12, 64).transpose(1, 2) for i in range(3))
    ref = (torch.softmax(q @ k.transpose(-1, -2), -1) @ v).transpose(1, 2).reshape(T, 768)
    err = float((out[:T].double().cpu() - ref).abs().max())
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.iters + 1)]
    ev[0].record()
    for i in range(a.iters):
        eng.diag_attention_bf16(x, B, T, q_has_log2e=True)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(a.iters))
    med = ms[len(ms) // 2]
    fl = 4.0 * B * 12 * T * T * 64
    print(json.dumps({"kernel": {"0": "v2 32x32x16, 32 queries / wave", "2": "v3 16x16x32, 32 queries / wave", "3": "v3 16x16x32, 32 queries / wave, builtin V reads", "4": "v3 16x16x32, 64 queries / wave, 4 waves", "8": "v3 16x16x32, 64 queries / wave, 8 waves", "16": "v3 16x16x32, 32 queries / wave, 16 waves", "5": "v3 16x16x32, 32 queries / wave, round 5 register use"}.get(os.environ.get("NOMAD_BF16_ATTN_V3", "2"), "?"), "B": B, "T": T,
                      "us_median": round(med * 1e3, 1), "us_min": round(ms[0] * 1e3, 1), "tflops": round(fl / med / 1e9, 1),
                      "max_abs_err_clip0": err}), flush=True)


if __name__ == "__main__":
    main()
