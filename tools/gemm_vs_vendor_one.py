#!/usr/bin/env python3
"""One shape through a nomad_diag_gemm tile and through torch.matmul (the vendor kernel), a few launches each: the target of
the rocprofv3 --pmc passes of tools/gpu_pmc_vendor.sh (counters per kernel name)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from nomad_amd.engine import Engine  # noqa: E402
from nomad_amd.weights import seeded_state_dict  # noqa: E402
from gemm_sweep import SHAPES  # noqa: E402

tiles = [int(t) for t in sys.argv[1].split(",")]
sname = sys.argv[2]
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
M, N, K, has_b, gelu, has_r = SHAPES[sname]
eng = Engine(seeded_state_dict(0), 0, diag=True)
g = torch.Generator().manual_seed(0)
A = torch.randn(M, K, generator=g).cuda()
W = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
b = torch.randn(N, generator=g).cuda() if has_b else None
R = torch.randn(M, N, generator=g).cuda() if has_r else None
C = torch.empty(M, N, device="cuda")
Wt = W.t()
for _ in range(iters):
    for t in tiles:
        if t < 0:
            torch.matmul(A, Wt, out=C)
        else:
            eng.diag_gemm(A, W, b, R, gelu=gelu, tile=t)
torch.cuda.synchronize()
print("done", tiles, sname)
