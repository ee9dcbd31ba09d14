#!/usr/bin/env python3
"""Alternating A/B of fp32 GEMM instantiations (nomad_diag_gemm tile ids) and the vendor library (tile -1 = torch.matmul, no
epilogue) on the hot shapes: every candidate is launched `--iters` times per round, the rounds alternate between the candidates
`--rounds` times, and the figure per candidate is the median over all its launches.  Output buffers are preallocated (nothing
but the kernel sits between two events); every candidate's result is compared bit for bit with the first one's.
Usage on the GPU box:  python3 tools/gemm_ab.py --tiles 33,65,-1 --shapes qkv,fc2 > gpurun_out/<tag>/gemm_ab.jsonl"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from nomad_amd import _lib  # noqa: E402
from nomad_amd.engine import Engine  # noqa: E402
from nomad_amd.weights import seeded_state_dict  # noqa: E402
from gemm_sweep import SHAPES  # noqa: E402

EXTRA = {  # quantisation probes for 256 x 128 tiles at 2 workgroups / CU (512 slots)
    "fc2_2r": (43520, 768, 3072, True, False, True),     # 170 x 6 = 1020 tiles: 1.99 rounds
    "fc2_3r": (65280, 768, 3072, True, False, True),     # 255 x 6 = 1530 tiles: 2.99 rounds
    "qkv_7r": (50944, 2304, 768, True, False, False),    # 199 x 18 = 3582 tiles: 7.0 rounds (= qkv)
    "conv2": (818944, 512, 1536, False, True, False),
    "fc2_h": (25472, 768, 3072, True, False, True),      # the bench batch as Engine.embed runs it: two halves on two streams
    "out_h": (25472, 768, 768, True, False, True),
    "fc1_h": (25472, 3072, 768, True, True, False),
    "qkv_h": (25472, 2304, 768, True, False, False),
    "conv6_h": (25472, 512, 1024, False, True, False),
    "c4_qkv": (1600, 2304, 768, True, False, False),     # configs[3]: 32 clips x T = 50 per branch
    "c4_out": (1600, 768, 768, True, False, True),
    "c4_fc1": (1600, 3072, 768, True, True, False),
    "c4_fc2": (1600, 768, 3072, True, False, True),
    "c4_conv1": (52384, 512, 1536, False, True, False),
    "c4_conv2": (26176, 512, 1536, False, True, False),
    "c4_conv3": (13056, 512, 1536, False, True, False),
    "c4_conv4": (6528, 512, 1536, False, True, False),
    "proj_h": (25472, 768, 512, True, False, False),
    "conv4": (204544, 512, 1536, False, True, False),
}
BNS = {67: 256}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", default="33,-1")
    ap.add_argument("--shapes", default="qkv,out,fc1,fc2,conv3")
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    eng = Engine(seeded_state_dict(0), 0, diag=True)
    g = torch.Generator().manual_seed(0)
    shapes = dict(SHAPES)
    shapes.update(EXTRA)
    tiles = [int(t) for t in a.tiles.split(",")]
    for sname in a.shapes.split(","):
        M, N, K, has_b, gelu, has_r = shapes[sname]
        A = torch.randn(M, K, generator=g).cuda()
        W = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
        b = torch.randn(N, generator=g).cuda() if has_b else None
        R = torch.randn(M, N, generator=g).cuda() if has_r else None
        Co = torch.empty(M, N, device="cuda")
        Wt = W.t()
        ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731

        def launch(tile):
            if tile < 0:
                torch.matmul(A, Wt, out=Co)
            else:
                _lib.check(eng.lib.nomad_diag_gemm(eng.ctx, ptr(A), ptr(W), ptr(b), ptr(R), ptr(Co), M, N, K, int(gelu), tile, eng._stream()), "diag_gemm")

        cand = [t for t in tiles if t < 0 or N % BNS.get(t % 100, 128) == 0]
        base, same, times = None, {}, {t: [] for t in cand}
        for t in list(cand):  # warm-up + bit check
            Co.zero_()
            try:
                launch(t)
            except Exception as e:  # noqa: BLE001 - a tile that does not take this shape (e.g. the direct epilogue with a residual)
                print(json.dumps({"shape": sname, "tile": t, "skipped": str(e)[-120:]}), flush=True)
                cand.remove(t)
                del times[t]
                continue
            torch.cuda.synchronize()
            if t >= 0:
                if base is None:
                    base = Co.clone()
                same[t] = bool(torch.equal(Co, base))
        for _ in range(a.rounds):
            for t in cand:
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.iters + 1)]
                launch(t)
                ev[0].record()
                for i in range(a.iters):
                    launch(t)
                    ev[i + 1].record()
                torch.cuda.synchronize()
                times[t] += [ev[i].elapsed_time(ev[i + 1]) for i in range(a.iters)]
        fl = 2.0 * M * N * K
        for t in cand:
            ts = sorted(times[t])
            med = ts[len(ts) // 2]
            print(json.dumps({"shape": sname, "M": M, "N": N, "K": K, "tile": t, "ms_med": round(med, 4), "ms_min": round(ts[0], 4),
                              "tflops": round(fl / med / 1e9, 1), "tflops_best": round(fl / ts[0] / 1e9, 1),
                              "bit_identical": same.get(t)}), flush=True)
        del A, W, b, R, Co, base
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
