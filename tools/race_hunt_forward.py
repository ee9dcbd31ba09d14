#!/usr/bin/env python3
"""Race hunt, fourth stage: a WHOLE forward (bf16, fp32 or bf16x3; one stream, no batch split) as the victim while one class
of kernels runs on another stream - does any kernel of the path lose bits to a co-runner, and (bf16: per-stage checksums)
which one?  With the library built WITH packed-FP32 instructions (NOMAD_LIB_VARIANT=pk) the bf16 128x128 GEMM aggressors make
the forward differ; the shipped build (no packed FP32) must show 0 everywhere.
Usage: [NOMAD_LIB_VARIANT=pk] python tools/race_hunt_forward.py [seconds per (victim, aggressor)] [victims] [aggressors]
victims: bf16,fp32,bf16x3; aggressors: none, bt1 / bt5 / bt11 / bt12 (bf16 128x128 GEMM variants), bt3, bt16, x3 (bf16x3 GEMM),
ft31 (fp32 GEMM), attn, forward16 (another bf16 forward), matmul (rocBLAS)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nomad_amd.engine import Engine
from nomad_amd.weights import num_frames, seeded_state_dict

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
victims = (sys.argv[2] if len(sys.argv) > 2 else "bf16,fp32,bf16x3").split(",")
aggressors = (sys.argv[3] if len(sys.argv) > 3 else "none,bt1,bt5,bt11,bt12,forward16").split(",")
eng = Engine(seeded_state_dict(0), 0, diag=True)
lib = eng.lib
lib.nomad_diag_set_cksum.restype = C.c_int
lib.nomad_diag_set_cksum.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
eng.F32_SPLIT_ROWS = eng.X3_SPLIT_ROWS = eng.BF16_SPLIT_ROWS = 0
gen = torch.Generator().manual_seed(33)
B, N = 32, 64000
wav_all = (0.1 * torch.randn(2 * B, N, generator=gen)).clamp(-1, 1).cuda()
wav = wav_all[:B].contiguous()
M = B * num_frames(N)
A768 = (torch.randn(M, 768, generator=gen) * 0.5).to(torch.bfloat16).cuda()
W768 = (torch.randn(768, 768, generator=gen) * 0.03).to(torch.bfloat16).cuda()
W2304 = (torch.randn(2304, 768, generator=gen) * 0.03).to(torch.bfloat16).cuda()
A1536 = (torch.randn(B * 6399 // 8, 1536, generator=gen) * 0.5).to(torch.bfloat16).cuda()
W512 = (torch.randn(512, 1536, generator=gen) * 0.03).to(torch.bfloat16).cuda()
A768f, W768f = A768.float(), W768.float()
qkv = (torch.randn(M, 2304, generator=gen) * 0.5).to(torch.bfloat16).cuda()
A3 = eng.diag_split_bf16(A768f)
W3 = eng.diag_split_bf16(W768f)
junk = torch.randn(4096, 4096, device="cuda")
emb_side = torch.empty(B, 256, device="cuda")
agg_stream = torch.cuda.Stream()
STAGES = (["gn_sums", "gn_scale", "gn_shift"] + [f"conv{i}" for i in range(7)] + ["feature_ln", "proj(xpad)", "posconv+res", "encoder_ln"]
          + [f"L{l}.{n}" for l in range(12) for n in ("qkv", "attn", "out_proj+res", "ln1", "fc1", "fc2+res", "ln2")] + ["emb"])
NST, SEGS = len(STAGES), 16 * B


def aggress(kind):
    global junk
    if kind == "none":
        return
    with torch.cuda.stream(agg_stream):
        if kind.startswith("bt"):
            t = int(kind[2:])
            if t == 3:
                eng.diag_gemm_bf16(A768, W2304, tile=3)
            elif t == 16:
                eng.diag_gemm_bf16(A1536, W512, tile=16)
            else:
                for _ in range(4):
                    eng.diag_gemm_bf16(A768, W768, tile=t)
        elif kind.startswith("ft"):
            eng.diag_gemm(A768f, W768f, tile=int(kind[2:]))
        elif kind == "x3":
            for _ in range(3):
                eng.diag_gemm_bf16x3(A3, W3)
        elif kind == "attn":
            for _ in range(3):
                eng.diag_attention_bf16(qkv, B, num_frames(N), True)
        elif kind == "forward16":
            eng._embed_bf16_into(wav_all[B:], emb_side, side=1)
        elif kind == "matmul":
            junk = junk @ junk * 1e-3


def victim(kind, tab=None):
    if kind == "bf16":
        if tab is not None:
            lib.nomad_diag_set_cksum(eng.ctx, tab.data_ptr(), NST, SEGS)
        return eng.embed_bf16(wav)
    return eng.embed(wav) if kind == "fp32" else eng.embed_bf16x3(wav)


lib.nomad_enable_bf16(eng.ctx)
lib.nomad_enable_bf16x3(eng.ctx)
variant = os.environ.get("NOMAD_LIB_VARIANT", "shipped (no packed FP32)")
for v in victims:
    reftab = torch.zeros(NST, SEGS, dtype=torch.int64, device="cuda")
    ref = victim(v, reftab).clone()
    torch.cuda.synchronize()
    for a in aggressors:
        t_end = time.time() + secs
        calls = bad = 0
        hist = {}
        while time.time() < t_end:
            for _ in range(4):
                aggress(a)
                tab = torch.zeros(NST, SEGS, dtype=torch.int64, device="cuda") if v == "bf16" else None
                out = victim(v, tab)
                calls += 1
                if not torch.equal(out, ref):
                    bad += 1
                    first = "?"
                    if tab is not None:
                        st_bad = torch.nonzero((tab != reftab).any(dim=1)).flatten().tolist()
                        first = STAGES[st_bad[0]] if st_bad else "emb only"
                    hist[first] = hist.get(first, 0) + 1
                    if bad <= 2:
                        rows = torch.nonzero((out != ref).any(dim=1)).flatten().tolist()
                        print(f"  [{v} vs {a}] call {calls}: {len(rows)} clips differ, max|diff| {(out - ref).abs().max().item():.3e}, first stage {first}", flush=True)
        torch.cuda.synchronize()
        print(f"library {variant}: victim {v} forward, aggressor {a}: mismatches {bad}/{calls}" + (f", first differing stage {hist}" if hist else ""), flush=True)
