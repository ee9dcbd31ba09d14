#!/usr/bin/env python3
"""Race screen for the deep-pipelined bf16 GEMM (gemm_bf16_8phase.hip.h): (1) the whole bf16 forward on 4 s and 30 s
batches, repeated - every run bit-identical; (2) tile 16 vs the older kernels on many shapes with a second stream
hammering HBM (changes DMA arrival order), outputs must agree exactly run to run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict
eng = Engine(seeded_state_dict(0), 0, diag=True)  # libnomad_diag.so: experimental tile ids
g = torch.Generator().manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
for B, ns in ((256, 64000), (32, 480000)):
    wav = (0.1 * torch.randn(B, ns, generator=g)).clamp(-1, 1).cuda()
    ref = eng.embed_bf16(wav).clone()
    for i in range(n // 2):
        bad += int(not torch.equal(eng.embed_bf16(wav), ref))
    del wav
side = torch.cuda.Stream()
junk = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
shapes = [(50944, 2304, 768), (47968, 768, 3072), (47968, 768, 768), (409344, 512, 1536), (12000, 3072, 768), (4096 + 17, 768, 512)]
for M, N, K in shapes:
    A = torch.randn(M, K, generator=g).bfloat16().cuda()
    W = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().cuda()
    b = torch.randn(N, generator=g).cuda()
    R = torch.randn(M, N, generator=g).bfloat16().cuda()
    ref = eng.diag_gemm_bf16(A, W, b, R, gelu=True, tile=16).clone()
    old = eng.diag_gemm_bf16(A, W, b, R, gelu=True, tile=1)
    d = (ref.float() - old.float()).abs().max().item() / old.float().abs().max().item()
    miss = 0
    for i in range(n):
        if i % 2:
            with torch.cuda.stream(side):
                junk.add_(1)  # background HBM traffic
        miss += int(not torch.equal(eng.diag_gemm_bf16(A, W, b, R, gelu=True, tile=16), ref))
    torch.cuda.synchronize()
    print(f"shape {M}x{N}x{K}: vs 128x128 kernel rel diff {d:.2e}, repeat mismatches {miss}/{n}")
    bad += miss + int(d > 1e-2)
print(f"soak_bf16: mismatches = {bad}")
sys.exit(1 if bad else 0)
