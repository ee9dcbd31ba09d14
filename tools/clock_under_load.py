#!/usr/bin/env python3
"""Shader clock while the production kernels run: a one-wave probe on a side stream counts shader cycles against the
100 MHz wall counter during (a) idle, (b) the fp32 GEMM (tile 33, fc1 shape), (c) the full fp32 forward, (d) the bf16
8-phase GEMM.  The MFMA peak a kernel can be held against scales with this clock."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict
eng = Engine(seeded_state_dict(0), 0, diag=True)  # libnomad_diag.so: experimental tile ids
side = torch.cuda.Stream()
g = torch.Generator().manual_seed(0)
A = torch.randn(50944, 768, generator=g).cuda(); W = (torch.randn(3072, 768, generator=g) * 0.03).cuda(); b = torch.randn(3072, generator=g).cuda()
A16, W16 = A.bfloat16(), W.bfloat16()
wav = (0.1 * torch.randn(256, 64000, generator=g)).clamp(-1, 1).cuda()
def load_gemm():
    for _ in range(40): eng.diag_gemm(A, W, b, None, gelu=True, tile=33)
def load_fwd():
    for _ in range(3): eng.embed(wav)
def load_bf16():
    for _ in range(150): eng.diag_gemm_bf16(A16, W16, b, None, gelu=True, tile=16)
Co = torch.empty(50944, 3072, device="cuda")
Wt = W.t()
def load_vendor():
    for _ in range(40): torch.matmul(A, Wt, out=Co)
def load_gemm_noepi():   # the shipped dispatch without bias / GELU (what the vendor line computes)
    for _ in range(40): eng.diag_gemm(A, W, None, None, gelu=False, tile=33)
qkv32 = (torch.randn(256 * 199, 2304, generator=g) * 0.5).cuda()
qkv16 = (torch.randn(32 * 1499, 2304, generator=g) * 0.5).cuda().bfloat16()
def load_attn32():
    for _ in range(60): eng.diag_attention(qkv32, 256, 199)
def load_attn16():
    for _ in range(100): eng.diag_attention_bf16(qkv16, 32, 1499, True)
res = {}
for name, fn, ms in (("fp32_attention_256x199", load_attn32, 20), ("bf16_attention_32x1499", load_attn16, 20), ("idle", lambda: None, 20), ("fp32_gemm_256x128_fc1", load_gemm, 40), ("fp32_gemm_256x128_fc1_no_epilogue", load_gemm_noepi, 40),
                     ("hipblaslt_fp32_fc1", load_vendor, 40), ("fp32_forward_256x4s", load_fwd, 200), ("bf16_gemm_8phase_fc1", load_bf16, 40)):
    fn(); torch.cuda.synchronize()          # warm
    fn()                                    # load in flight on the main stream
    out = eng.diag_clock_probe(ms, side)
    fn()
    torch.cuda.synchronize()
    cyc, ticks = out.tolist()
    res[name] = round(cyc / (ticks / 100.0), 1)
    print(f"{name:28s} shader clock {res[name]:7.1f} MHz over {ticks / 1e5:.1f} ms")
print(json.dumps(res))
