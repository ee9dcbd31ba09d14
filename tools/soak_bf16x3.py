#!/usr/bin/env python3
"""Race screen for the bf16x3 path (gemm_bf16x3.hip.h, attention_x3_kernel, the Toeplitz pos-conv): (1) the whole
forward on 4 s, 30 s and ragged batches, repeated - every run bit-identical; (2) the staged-once GEMM against the
K-concatenated one on the model's shapes with a second stream hammering HBM (changes DMA arrival order): the two kernels
add the same products in a different order (results within fp32 rounding of each other), each is bit-stable run to run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict
eng = Engine(seeded_state_dict(0), 0, diag=True)  # libnomad_diag.so: experimental tile ids
g = torch.Generator().manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
for B, ns in ((256, 64000), (32, 480000), (7, 100001)):
    wav = (0.1 * torch.randn(B, ns, generator=g)).clamp(-1, 1).cuda()
    ref = eng.embed_bf16x3(wav).clone()
    miss = sum(int(not torch.equal(eng.embed_bf16x3(wav), ref)) for _ in range(n // 2))
    print(f"forward {B} x {ns}: repeat mismatches {miss}/{n // 2}")
    bad += miss
    del wav
lens = torch.randint(400, 8 * 16000, (200,), generator=g).tolist()
waves = [(0.1 * torch.randn(k, generator=g)).clamp(-1, 1).cuda() for k in lens]
ref = eng.embed_ragged(waves, precision="bf16x3").clone()
miss = sum(int(not torch.equal(eng.embed_ragged(waves, precision="bf16x3"), ref)) for _ in range(n // 2))
print(f"ragged 200 clips: repeat mismatches {miss}/{n // 2}")
bad += miss
side = torch.cuda.Stream()
junk = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
shapes = [(50944, 2304, 768), (47968, 768, 3072), (47968, 768, 768), (409344, 512, 1536), (12000, 3072, 768), (4096 + 17, 768, 512)]
for M, N, K in shapes:
    A = eng.diag_split_bf16(torch.randn(M, K, generator=g).cuda())
    W = eng.diag_split_bf16((torch.randn(N, K, generator=g) * K ** -0.5).cuda())
    b = torch.randn(N, generator=g).cuda()
    R = eng.diag_split_bf16(torch.randn(M, N, generator=g).cuda())
    ref = eng.diag_gemm_bf16x3(A, W, b, R, gelu=True, variant=8).clone()
    old = eng.diag_gemm_bf16x3(A, W, b, R, gelu=True, variant=1)
    d = (ref - old).abs().max().item() / old.abs().max().item()
    miss = 0
    for i in range(n):
        if i % 2:
            with torch.cuda.stream(side):
                junk.add_(1)  # background HBM traffic
        miss += int(not torch.equal(eng.diag_gemm_bf16x3(A, W, b, R, gelu=True, variant=8), ref))
    torch.cuda.synchronize()
    print(f"shape {M}x{N}x{K}: staged vs K-concatenated rel diff {d:.2e}, repeat mismatches {miss}/{n}")
    bad += miss + int(d > 1e-5)
print(f"soak_bf16x3: mismatches = {bad}")
sys.exit(1 if bad else 0)
