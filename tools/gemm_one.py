#!/usr/bin/env python3
"""Run ONE GEMM instantiation on one hot shape a few times (target for rocprofv3 --pmc passes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from nomad_amd.engine import Engine  # noqa: E402
from nomad_amd.weights import seeded_state_dict  # noqa: E402
from gemm_sweep import SHAPES  # noqa: E402

tile = int(sys.argv[1]) if len(sys.argv) > 1 else 0
sname = sys.argv[2] if len(sys.argv) > 2 else "fc1"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
M, N, K, has_b, gelu, has_r = SHAPES[sname]
eng = Engine(seeded_state_dict(0), 0, diag=True)  # libnomad_diag.so: experimental tile ids
if os.environ.get("GEMM_X3") == "1":   # bf16x3 products on fp32 buffers (gemm_f32_glds_kernel<..., X3>)
    eng.gemm_precision = "bf16x3"
g = torch.Generator().manual_seed(0)
A = torch.randn(M, K, generator=g).cuda()
W = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
b = torch.randn(N, generator=g).cuda() if has_b else None
R = torch.randn(M, N, generator=g).cuda() if has_r else None
for _ in range(iters):
    eng.diag_gemm(A, W, b, R, gelu=gelu, tile=tile)
torch.cuda.synchronize()
print("done", tile, sname)
