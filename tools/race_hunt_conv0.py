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
lib.nomad_diag_conv0_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
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
    rc = lib.nomad_diag_conv0_bf16(eng.ctx, w.data_ptr(), w.shape[0], N, out.data_ptr(), scratch.data_ptr(), torch.cuda.current_stream().cuda_stream)
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
    torch.cuda.synchronize()
    print(f"aggressor {kind}: conv0 mismatches {bad}/{calls}", flush=True)
