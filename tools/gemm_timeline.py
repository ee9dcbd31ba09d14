#!/usr/bin/env python3
"""Where a 256x256 bf16 GEMM tile's time goes on its CU: per-workgroup wall-clock stamps (entry, main loop start, main loop
end, output stores done) from the 8-phase kernel's timing-probe instantiation (libnomad_diag.so, tile id 36), grouped by CU.
Usage: python tools/gemm_timeline.py [--shapes c5_qkv,c5_k128]"""
import argparse, collections, ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict
from gemm_sweep import SHAPES


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="c5_qkv,c5_k128,c5_fc2")
    a = ap.parse_args()
    eng = Engine(seeded_state_dict(0), 0, diag=True)
    g = torch.Generator().manual_seed(0)
    for sname in a.shapes.split(","):
        M, N, K, has_b, gelu, has_r = SHAPES[sname]
        A = torch.randn(M, K, generator=g).cuda().bfloat16()
        W = (torch.randn(N, K, generator=g) * K ** -0.5).cuda().bfloat16()
        b = torch.randn(N, generator=g).cuda() if has_b else None
        R = torch.randn(M, N, generator=g).cuda().bfloat16() if has_r else None
        for _ in range(5):
            out = eng.diag_gemm_bf16(A, W, b, R, gelu=gelu, tile=36)
        torch.cuda.synchronize()
        nwg = ((M + 255) // 256) * (N // 256)
        n = min(nwg, 4096)
        buf = (C.c_ulonglong * (6 * n))()
        rc = eng.lib.nomad_diag_timeline(buf, n)
        assert rc == 0
        t = np.frombuffer(buf, dtype=np.uint64).reshape(n, 6).astype(np.int64)
        t0 = t[:, 0].min()
        us = (t[:, :4] - t0) / 100.0                       # 100 MHz ticks -> us
        pro, loop, epi = us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2]
        hw, xcc = t[:, 4], t[:, 5] & 0xF
        cu = (xcc << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 8) | ((hw >> 8) & 0xF)   # (xcc, se, sh, cu)
        per_cu = collections.defaultdict(list)
        for i in range(n):
            per_cu[int(cu[i])].append(i)
        gaps = []
        for ids in per_cu.values():
            ids.sort(key=lambda i: us[i, 0])
            gaps += [us[b2, 0] - us[a2, 3] for a2, b2 in zip(ids, ids[1:])]
        gaps = gaps or [0.0]
        first = us[:, 0] < 1.0
        res = {"shape": sname, "M": M, "N": N, "K": K, "workgroups": nwg, "distinct_cus": len(per_cu),
               "kernel_span_us": round(float(us[:, 3].max()), 1),
               "entry_to_loop_us": {"first_round": round(float(pro[first].mean()), 2), "later_rounds": round(float(pro[~first].mean()), 2) if (~first).any() else None},
               "main_loop_us": round(float(loop.mean()), 2), "epilogue_us": round(float(epi.mean()), 2),
               "gap_between_workgroups_on_a_cu_us": {"mean": round(float(np.mean(gaps)), 2), "median": round(float(np.median(gaps)), 2),
                                                      "p90": round(float(np.percentile(gaps, 90)), 2)},
               "workgroups_per_cu": round(n / len(per_cu), 2)}
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
