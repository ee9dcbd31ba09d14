#!/usr/bin/env python3
"""Burst vs sustained rate of one bf16 GEMM instantiation: the same launch repeated for ~1.5 s (long enough for the chip's clock
management to settle) with the shader clock read by the one-wave probe on a side stream, next to the median of a burst of 8.
Usage: python3 tools/sustained_gemm.py [shape,...] [tile,...]      (tile -1 = torch.matmul, i.e. hipBLASLt)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nomad_amd import _lib
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict
from gemm_sweep import SHAPES
shapes = (sys.argv[1] if len(sys.argv) > 1 else "c5_fc2,c5_out,c5_qkv").split(",")
tiles = [int(t) for t in (sys.argv[2] if len(sys.argv) > 2 else "16,55").split(",")]
eng = Engine(seeded_state_dict(0), 0, diag=True)
side = torch.cuda.Stream()
g = torch.Generator().manual_seed(0)
ptr = lambda t: t.data_ptr() if t is not None else None
for name in shapes:
    M, N, K, has_b, gelu, has_r = SHAPES[name]
    A = torch.randn(M, K, generator=g).cuda().bfloat16()
    W = (torch.randn(N, K, generator=g) * K ** -0.5).cuda().bfloat16()
    b = torch.randn(N, generator=g).cuda() if has_b else None
    R = torch.randn(M, N, generator=g).cuda().bfloat16() if has_r else None
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for tile in tiles:
        if tile in (55, 56) and N % 192:
            continue
        if tile == -1:   # the vendor library on the same operands (plain A W^T)
            Wt = W.t()
            fn = lambda: torch.matmul(A, Wt, out=C)
        else:
            fn = lambda: _lib.check(eng.lib.nomad_diag_gemm_bf16(eng.ctx, ptr(A), ptr(W), ptr(b), ptr(R), ptr(C), M, N, K, int(gelu), tile, eng._stream()), "gemm")
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        import time; time.sleep(0.5)     # let the chip idle: the burst starts from a cool state
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
        ev[0].record()
        for i in range(8):
            fn(); ev[i + 1].record()
        torch.cuda.synchronize()
        burst = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(8))[4]
        n = max(200, int(1500.0 / burst))
        for _ in range(n // 4):
            fn()                                   # ramp
        probe = eng.diag_clock_probe(min(800.0, 0.6 * n * burst), side)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        cyc, ticks = probe.tolist()
        sus = e0.elapsed_time(e1) / n
        fl = 2.0 * M * N * K
        print(json.dumps({"shape": name, "tile": tile, "burst_us": round(burst * 1e3, 1), "burst_tflops": round(fl / burst / 1e9, 1),
                          "sustained_us": round(sus * 1e3, 1), "sustained_tflops": round(fl / sus / 1e9, 1), "launches": n,
                          "shader_mhz_sustained": round(cyc / (ticks / 100.0), 1)}), flush=True)
