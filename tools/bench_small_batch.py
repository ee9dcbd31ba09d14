#!/usr/bin/env python3
"""Which path serves a small batch fastest?  embed() on fp32 buffers with exact fp32 MFMA GEMMs, the same with bf16x3 products
(Engine.gemm_precision = "bf16x3": hi / lo split in registers, same small tiles), and embed_bf16x3 (split storage, 256 x 256
tiles) over batch sizes and clip lengths - sets Nomad's BF16X3_MIN_SAMPLES.  Usage: python tools/bench_small_batch.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nomad_amd.engine import Engine
from nomad_amd.weights import num_frames, seeded_state_dict

eng = Engine(seeded_state_dict(0), 0)
gen = torch.Generator().manual_seed(0)


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


for N in (16384, 64000, 160000):
    for B in (1, 2, 4, 8, 16, 32, 64, 128, 256):
        if B * N > 256 * 64000:
            continue
        wav = (0.1 * torch.randn(B, N, generator=gen)).clamp(-1, 1).cuda()
        eng.gemm_precision = "fp32"
        t_f32 = timed(lambda: eng.embed(wav))
        ref = eng.embed(wav)
        eng.gemm_precision = "bf16x3"
        t_x3f = timed(lambda: eng.embed(wav))
        err = (eng.embed(wav) - ref).abs().max().item()
        eng.gemm_precision = "fp32"
        t_x3s = timed(lambda: eng.embed_bf16x3(wav))
        print(json.dumps({"clips": B, "samples": N, "rows": B * num_frames(N), "fp32_ms": round(t_f32, 3), "x3_products_fp32_buffers_ms": round(t_x3f, 3),
                          "x3_split_storage_ms": round(t_x3s, 3), "emb_max_abs_diff_x3_products_vs_fp32": err}), flush=True)
