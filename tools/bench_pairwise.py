#!/usr/bin/env python3
"""The distance stage alone at config C3's size: 10 000 degraded x 1 000 reference embeddings -> float64 distance matrix
(80 MB written) + row means.  Reports device time per launch (hipEvents on the launch stream) and GB/s of the matrix
write, next to the ~5-6.3 TB/s a streaming write reaches on MI355X.  Usage: python tools/bench_pairwise.py [Nd] [Nr]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from nomad_amd.engine import Engine  # noqa: E402
from nomad_amd.weights import seeded_state_dict  # noqa: E402

Nd = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
Nr = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
eng = Engine(seeded_state_dict(0), 0)
g = torch.Generator().manual_seed(0)
deg = torch.nn.functional.normalize(torch.randn(Nd, 256, generator=g), dim=1).cuda()
ref = torch.nn.functional.normalize(torch.randn(Nr, 256, generator=g), dim=1).cuda()
out = {"Nd": Nd, "Nr": Nr}
for name, want in (("matrix_and_means", True), ("means_only", False)):
    for _ in range(3):
        eng.pairwise(deg, ref, want_matrix=want)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        d, m = eng.pairwise(deg, ref, want_matrix=want)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    out[name] = {"ms_per_launch": round(ms, 4), "pairs_per_s": round(Nd * Nr / ms * 1e3, 1),
                 "write_GB_per_s": round(Nd * Nr * 8 / ms / 1e6, 1) if want else None,
                 "fp64_lane_ops_per_s": round(Nd * Nr * 256 * 2 / ms * 1e3, 1)}
import scipy.spatial.distance as sd
d_ref = sd.cdist(deg[:64].cpu().numpy(), ref.cpu().numpy())
out["max_abs_err_vs_scipy_64_rows"] = float(abs(d[:64].cpu().numpy() - d_ref).max()) if d is not None else None
d, m = eng.pairwise(deg, ref, want_matrix=True)
out["max_abs_err_vs_scipy_64_rows"] = float(abs(d[:64].cpu().numpy() - d_ref).max())
print(json.dumps(out))
