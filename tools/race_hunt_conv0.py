#!/usr/bin/env python3
"""Race hunt, third stage: the bf16 forward's front end ALONE (wav statistics -> GroupNorm fold -> conv0 + GroupNorm + GELU,
nomad_diag_conv0_bf16) as the victim on one stream, one class of kernels as the aggressor on another, in a tight loop:
which co-runner makes conv0's output differ from its reference, and HOW does it differ (which clips / frames / channels,
by how much)?
Usage: python tools/race_hunt_conv0.py [seconds per aggressor] [aggressors, comma separated]
aggressors: none, forward (the other half's whole bf16 forward), gemm1 / gemm3 / gemm16 / gemm2 (bf16 GEMM tiles 128x128,
256x256, 8-phase, 128x64), attn (bf16 attention), matmul (rocBLAS fp32), f32 (the fp32 forward), conv0 (a second conv0)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nomad_amd.engine import Engine
from nomad_amd.weights import num_frames, seeded_state_dict

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
which = (sys.argv[2] if len(sys.argv) > 2 else "none,forward,gemm1,gemm3,gemm16,gemm2,attn,matmul,f32,conv0").split(",")
eng = Engine(seeded_state_dict(0), 0, diag=True)
lib = eng.lib
lib.nomad_diag_conv0_bf16.restype = C.c_int
lib.nomad_diag_conv0_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
VARIANT = int(os.environ.get("HUNT_VARIANT", 0))   # conv0_gn_gelu_kernel's VAR: bit 0 no LDS, bit 1 scalar tap loop (no v_pk_fma_f32)
lib.nomad_enable_bf16(eng.ctx)
gen = torch.Generator().manual_seed(33)
if os.environ.get("HUNT_PARTS"):   # prior activity in the process, as in tools/race_hunt_stages.py ("d": degenerate inputs)
    for kind in (torch.zeros(2, 16000), torch.full((2, 16000), 0.5), torch.ones(2, 16000)):
        eng.embed(kind.cuda())
B, N = 32, 64000
L0 = (N - 10) // 5 + 1
wav_all = (0.1 * torch.randn(2 * B, N, generator=gen)).clamp(-1, 1).cuda()
wav = wav_all[:B]


def conv0(w, out, scratch):
    rc = lib.nomad_diag_conv0_bf16(eng.ctx, w.data_ptr(), w.shape[0], N, out.data_ptr(), scratch.data_ptr(), torch.cuda.current_stream().cuda_stream, VARIANT)
    assert rc == 0


out = torch.empty(B, L0, 512, dtype=torch.bfloat16, device="cuda")
scratch = torch.empty(8 * 65 * B * 4 + 8 * 512 * B + 4096, dtype=torch.uint8, device="cuda")
conv0(wav, out, scratch)
torch.cuda.synchronize()
ref = out.clone()
ref_i = ref.view(torch.int16)
agg_stream = torch.cuda.Stream()
# aggressor operands
M = B * num_frames(N)
A768 = (torch.randn(M, 768, generator=gen) * 0.5).to(torch.bfloat16).cuda()
W2304 = (torch.randn(2304, 768, generator=gen) * 0.03).to(torch.bfloat16).cuda()
W768 = (torch.randn(768, 768, generator=gen) * 0.03).to(torch.bfloat16).cuda()
A1536 = (torch.randn(B * 6399 // 8, 1536, generator=gen) * 0.5).to(torch.bfloat16).cuda()
W512 = (torch.randn(512, 1536, generator=gen) * 0.03).to(torch.bfloat16).cuda()
qkv = (torch.randn(M, 2304, generator=gen) * 0.5).to(torch.bfloat16).cuda()
A768f, W768f = A768.float(), W768.float()
junk = torch.randn(4096, 4096, device="cuda")
out2 = torch.empty_like(out)
scratch2 = torch.empty_like(scratch)
emb_side = torch.empty(B, 256, device="cuda")


def aggress(kind):
    global junk
    if kind == "none":
        return
    with torch.cuda.stream(agg_stream):
        if kind == "forward":
            eng._embed_bf16_into(wav_all[B:], emb_side, side=1)
        elif kind == "f32":
            eng.embed(wav_all[B:], side=True)
        elif kind == "gemm1":
            for _ in range(6):
                eng.diag_gemm_bf16(A768, W768, tile=1)
        elif kind.startswith("bt"):      # bt<N>: bf16 GEMM diag tile N on the N = 768 shape (0 256x128, 4 64x64, 5 128x128 4 waves,
            for _ in range(6):           # 6 256x128 4 waves, 7 = gemm1 without its stores, 8 = gemm1 with one K tile, 11 / 12 BK 32 / 3-stage)
                eng.diag_gemm_bf16(A768, W768, tile=int(kind[2:]))
        elif kind.startswith("ft"):      # ft<N>: fp32 GEMM diag tile N (31 = 128x128x32 8 waves, 20 = 4 waves, 37 = 64x64, 34 = 128x64)
            for _ in range(2):
                eng.diag_gemm(A768f, W768f, tile=int(kind[2:]))
        elif kind == "gemm3":
            for _ in range(3):
                eng.diag_gemm_bf16(A768, W2304, tile=3)
        elif kind == "gemm16":
            for _ in range(3):
                eng.diag_gemm_bf16(A1536, W512, tile=16)
        elif kind == "gemm2":
            for _ in range(6):
                eng.diag_gemm_bf16(A768, W768[:64 * 11], tile=2)
        elif kind == "attn":
            for _ in range(4):
                eng.diag_attention_bf16(qkv, B, num_frames(N), True)
        elif kind == "matmul":
            for _ in range(2):
                junk = junk @ junk * 1e-3
        elif kind == "conv0":
            for _ in range(3):
                conv0(wav_all[B:], out2, scratch2)


def describe(o):
    d = (o.view(torch.int16) != ref_i)
    n = int(d.sum())
    clips = torch.nonzero(d.flatten(1).any(dim=1)).flatten().tolist()
    frames = d.any(dim=2)                              # (B, L0)
    per_clip = frames.sum(dim=1).tolist()
    b0 = clips[0]
    fr = torch.nonzero(frames[b0]).flatten().tolist()
    ch = torch.nonzero(d[b0, fr[0]]).flatten().tolist()
    diff = (o.float() - ref.float()).abs()
    rel = (diff / ref.float().abs().clamp_min(1e-3))[d]
    print(f"      {n} elements differ in clips {clips[:6]}{'...' if len(clips) > 6 else ''} ({len(clips)}); frames per clip "
          f"{[per_clip[c] for c in clips[:6]]}; clip {b0}: frames {fr[:10]}{'...' if len(fr) > 10 else ''}, frame {fr[0]}: "
          f"{len(ch)} channels {ch[:12]}{'...' if len(ch) > 12 else ''}; max|diff| {float(diff.max()):.3e}, relative diff "
          f"median {float(rel.median()):.2e} max {float(rel.max()):.2e}; zeros among the differing: {int((o[d] == 0).sum())}", flush=True)


def gelu64(z):
    import math
    return 0.5 * z * (1.0 + math.erf(z / math.sqrt(2.0)))


def provenance(o, limit=6):
    """For a few differing elements: invert the GELU (z >= -0.75 branch) to the conv sum y the kernel must have had, and list
    A(j0) = y_obs - sum_{j > j0} w_j x_j for every j0 - the value the accumulator held after tap j0 IF the taps after it were
    applied correctly - next to the ten samples, to see what replaced the accumulator."""
    import numpy as np
    sd = eng._state_dict
    w0 = sd["ssl_model.feature_extractor.conv_layers.0.0.weight"].double().reshape(512, 10).numpy()
    sc_sh = scratch.cpu()
    nst = 65 * B * (1 + (L0 + 8191) // 8192)
    f = sc_sh[8 * nst:8 * nst + 8 * 512 * B].view(torch.float32).double().numpy()
    scale, shift = f[:512 * B].reshape(B, 512), f[512 * B:].reshape(B, 512)
    d = torch.nonzero(o.view(torch.int16) != ref_i)[:limit].tolist()
    wv = wav.double().cpu().numpy()
    for b, t, c in d:
        x = wv[b, 5 * t:5 * t + 10]
        y_ref = float((w0[c] * x).sum())
        o_obs, o_ref = float(o[b, t, c].float()), float(ref[b, t, c].float())
        lo, hi = -0.75, 30.0
        for _ in range(80):
            mid = 0.5 * (lo + hi)
            if gelu64(mid) < o_obs:
                lo = mid
            else:
                hi = mid
        z = 0.5 * (lo + hi)
        y_obs = (z - shift[b, c]) / scale[b, c]
        tail = np.cumsum((w0[c] * x)[::-1])[::-1]          # tail[j] = sum_{k >= j} w_k x_k
        A = [y_obs - (tail[j0 + 1] if j0 + 1 < 10 else 0.0) for j0 in range(-1, 10)]
        print(f"      elem (clip {b}, frame {t}, ch {c} = lane {(c // 4) % 64} q {c % 4}): out {o_obs:+.4f} ref {o_ref:+.4f}; y_ref {y_ref:+.5f} y_obs~{y_obs:+.5f} "
              f"(scale {scale[b, c]:.3f} shift {shift[b, c]:+.3f})\n        A(j0=-1..9) = {[round(a, 4) for a in A]}\n        x = {[round(float(v), 4) for v in x]}\n"
              f"        w = {[round(float(v), 4) for v in w0[c]]}", flush=True)


for kind in which:
    t_end = time.time() + secs
    calls = bad = 0
    while time.time() < t_end:
        for _ in range(8):
            aggress(kind)
            conv0(wav, out, scratch)
            same = torch.equal(out.view(torch.int16), ref_i)     # synchronises this stream only
            calls += 1
            if not same:
                bad += 1
                if bad <= 3:
                    print(f"  [{kind}] call {calls}: conv0 output differs", flush=True)
                    describe(out)
                    if bad <= 2 and os.environ.get("HUNT_PROVENANCE", "1") == "1":
                        provenance(out)
    torch.cuda.synchronize()
    print(f"variant {VARIANT} aggressor {kind}: conv0 mismatches {bad}/{calls}", flush=True)
