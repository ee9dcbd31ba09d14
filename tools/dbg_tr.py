"""Bit comparison of experimental fp32 GEMM tile ids against the production tile 33 on a few shapes (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict
eng = Engine(seeded_state_dict(0), 0, diag=True)
g = torch.Generator().manual_seed(0)
tiles = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "72,73,77,82").split(",")]
for (M, N, K, hb, gelu, hr) in [(512, 256, 64, True, False, False), (512, 256, 64, True, True, False), (512, 256, 64, True, False, True),
                                (1000, 384, 48, False, False, False), (1000, 384, 64, True, False, True), (50944, 768, 768, True, False, True), (70000, 128, 32, True, False, True), (333333, 512, 96, False, True, False),
                                (50944, 2304, 768, True, False, False), (50944, 768, 3072, True, False, True),
                                (25472, 768, 3072, True, False, True), (50944, 3072, 768, True, True, False), (25400, 768, 768, True, False, True)]:
    A = torch.randn(M, K, generator=g).cuda(); W = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
    b = torch.randn(N, generator=g).cuda() if hb else None
    R = torch.randn(M, N, generator=g).cuda() if hr else None
    base = eng.diag_gemm(A, W, b, R, gelu=gelu, tile=33)
    for t in tiles:
        try:
            o = eng.diag_gemm(A, W, b, R, gelu=gelu, tile=t)
        except Exception as e:  # noqa: BLE001
            print(M, N, K, "tile", t, "skipped:", str(e)[-100:], flush=True)
            continue
        d = (o - base)
        bad = (d != 0).nonzero()
        print(M, N, K, "gelu" if gelu else "-", "R" if hr else "-", "tile", t, "ndiff", bad.shape[0], "maxabs", float(d.abs().max()), "nan", int(torch.isnan(o).sum()),
              "first", bad[:4].tolist(), flush=True)
