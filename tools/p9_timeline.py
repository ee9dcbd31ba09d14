#!/usr/bin/env python3
"""Round 5: per-workgroup stamps of the persistent bf16 GEMM (gemm_bf16_p9.hip.h, probe instantiation = diag tile 61):
entry -> first K loop start -> K loop end of tile 0 -> epilogue end of tile 0 (= K loop start of tile 1) -> K loop end of tile 1.
Usage: python tools/p9_timeline.py [--shapes c5_qkv,c5_fc1,c5_fc2,c5_out]"""
import argparse, ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict
from gemm_sweep import SHAPES


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="c5_qkv,c5_fc1,c5_fc2,c5_out,c5_conv4")
    a = ap.parse_args()
    eng = Engine(seeded_state_dict(0), 0, diag=True)
    g = torch.Generator().manual_seed(0)
    for sname in a.shapes.split(","):
        M, N, K, has_b, gelu, has_r = SHAPES[sname]
        A = torch.randn(M, K, generator=g).cuda().bfloat16()
        W = (torch.randn(N, K, generator=g) * K ** -0.5).cuda().bfloat16()
        b = torch.randn(N, generator=g).cuda() if has_b else None
        R = torch.randn(M, N, generator=g).cuda().bfloat16() if has_r else None
        out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        for _ in range(5):
            eng.diag_gemm_bf16(A, W, b, R, gelu=gelu, tile=61, out=out)
        torch.cuda.synchronize()
        n = 256
        buf = (C.c_ulonglong * (6 * n))()
        assert eng.lib.nomad_diag_timeline(buf, n) == 0
        t = np.frombuffer(buf, dtype=np.uint64).reshape(n, 6).astype(np.int64)
        done = t[:, 5] >> 32
        t0 = t[:, 0].min()
        us = (t[:, :5] - t0) / 100.0
        two = done >= 2
        res = {"shape": sname, "M": M, "N": N, "K": K, "k_tiles": K // 64, "tiles": ((M + 255) // 256) * (N // 256),
               "workgroups_with_2plus_tiles": int(two.sum()),
               "entry_to_first_loop_us": round(float((us[:, 1] - us[:, 0]).mean()), 2),
               "k_loop_tile0_us": round(float((us[:, 2] - us[:, 1]).mean()), 2),
               "epilogue_tile0_us": round(float((us[:, 3] - us[:, 2]).mean()), 2),
               "k_loop_tile1_us": round(float((us[two, 4] - us[two, 3]).mean()), 2) if two.any() else None,
               "per_k_tile_us_tile1": round(float((us[two, 4] - us[two, 3]).mean()) / (K // 64), 3) if two.any() else None,
               "epilogue_tile0_us_median": round(float(np.median(us[:, 3] - us[:, 2])), 2),
               "k_loop_tile1_us_median": round(float(np.median(us[two, 4] - us[two, 3])), 2) if two.any() else None,
               "start_spread_us": round(float(us[:, 0].max()), 2),
               "skew_x10ns": int(os.environ.get("NOMAD_BF16_P9_SKEW", "0")),
               "tiles_done_max": int(done.max())}
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
