#!/usr/bin/env python3
"""Time the fp32 MFMA GEMM instantiations (nomad_diag_gemm tile ids) on the hot shapes of config C2.
Usage on the GPU box:  python3 tools/gemm_sweep.py [--tiles 0,3,4] [--shapes fc1,qkv] [--iters 5]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from nomad_amd.engine import Engine  # noqa: E402
from nomad_amd.weights import seeded_state_dict  # noqa: E402

SHAPES = {  # name: (M, N, K, bias, gelu, residual)
    "out": (50944, 768, 768, True, False, True),
    "qkv": (50944, 2304, 768, True, False, False),
    "fc1": (50944, 3072, 768, True, True, False),
    "fc2": (50944, 768, 3072, True, False, True),
    "conv3": (409344, 512, 1536, False, True, False),
    # half of the bench batch: what Engine.embed launches (two halves on two streams)
    "out_h": (25472, 768, 768, True, False, True),
    "qkv_h": (25472, 2304, 768, True, False, False),
    "fc1_h": (25472, 3072, 768, True, True, False),
    "fc2_h": (25472, 768, 3072, True, False, True),
    "conv3_h": (204672, 512, 1536, False, True, False),
    "conv1": (1638144, 512, 1536, False, True, False),   # dense stand-in for conv1's implicit GEMM (10 GB of split A)
    "conv5": (102144, 512, 1024, False, True, False),
    "conv6": (50944, 512, 1024, False, True, False),
    "fc1_nogelu": (50944, 3072, 768, True, False, False),
    "proj": (50944, 768, 512, True, False, False),
    "pos": (50944, 48, 6144, True, True, False),   # one group of the pos-conv (tile ids 48 / 49 only)
    # quantisation probes for the 256x128 kernel at 2 workgroups/CU (512 slots): 9.0, 9.33 and 9.98 rounds
    "fc1_9r": (49152, 3072, 768, True, True, False),
    "fc1_10r": (54528, 3072, 768, True, True, False),
    "c4_qkv": (1600, 2304, 768, True, False, False),
    "c4_fc1": (1600, 3072, 768, True, True, False),
    "c4_fc2": (1600, 768, 3072, True, False, True),
    "c4_out": (1600, 768, 768, True, False, True),
    "b1_qkv": (199, 2304, 768, True, False, False),
    "b1_fc2": (199, 768, 3072, True, False, True),
    "b8_fc1": (1592, 3072, 768, True, True, False),
    # one branch of the reference's training batch: 8 clips x 10 s (T = 499)
    "tr_qkv": (3992, 2304, 768, True, False, False),
    "tr_fc1": (3992, 3072, 768, True, True, False),
    "tr_fc2": (3992, 768, 3072, True, False, True),
    "tr_out": (3992, 768, 768, True, False, True),
    # the merged training batch: 3 x 8 clips x 10 s
    "tm_qkv": (11976, 2304, 768, True, False, False),
    "tm_fc1": (11976, 3072, 768, True, True, False),
    "tm_fc2": (11976, 768, 3072, True, False, True),
    "tm_out": (11976, 768, 768, True, False, True),
    "tm_dxin": (11976, 768, 2304, False, False, True),
    # weight-gradient shapes (rows = out features, cols = in features, contraction = padded rows / split)
    "dw_fc1": (3072, 768, 2048, False, False, False),
    "dw_out": (768, 768, 512, False, False, False),
    # config C5 (32 clips x 30 s, T = 1499): the encoder GEMMs and two conv layers
    "c5_out": (47968, 768, 768, True, False, True),
    "c5_out_nores": (47968, 768, 768, True, False, False),            # what the residual epilogue (between tiles) costs against the interleaved one
    "c5_fc2_nores": (47968, 768, 3072, True, False, False),
    "c5_qkv": (47968, 2304, 768, True, False, False),
    "c5_fc1": (47968, 3072, 768, True, True, False),
    "c5_fc2": (47968, 768, 3072, True, False, True),
    "c5h_out": (23984, 768, 768, True, False, True),      # one half of the two-stream split of config C5
    "c5h_fc2": (23984, 768, 3072, True, False, True),
    "c5_fc1_nogelu": (47968, 3072, 768, True, False, False),          # what the GELU epilogue costs on the bf16 tiles
    "c5_conv4_nogelu": (383968, 512, 1536, False, False, False),
    "one_tile": (256, 256, 128, True, False, False),        # a lone workgroup / one workgroup per XCD / one full round
    "eight_tiles": (256, 2048, 128, True, False, False),
    "one_round": (7168, 2304, 128, True, False, False),
    "c5_k128": (47968, 2304, 128, True, False, False),      # per-tile constants: the same tiles with 2 / 4 / 8 K tiles only
    "c5_k256": (47968, 2304, 256, True, False, False),
    "c5_k512": (47968, 2304, 512, True, False, False),
    "c5_conv2": (1535968, 512, 1536, False, True, False),
    "c5_conv4": (383968, 512, 1536, False, True, False),
    # large squares, for comparison with the guide's numbers for the 256x256 8-phase template (1320-1470 TFLOP/s)
    "sq4096": (4096, 4096, 4096, False, False, False),
    "sq8192": (8192, 8192, 8192, False, False, False),
}
TILE_NAMES = {0: "128x128x32 w2x2", 1: "128x64x16 w2x2", 2: "64x64x32 w2x2", 3: "128x128x16 w2x2",
              4: "256x128x32 w4x2", 5: "256x256x32 w4x2", 6: "256x128x16 w4x2", 7: "128x256x32 w2x2",
              8: "256x256x16 w4x2", 9: "128x128x16 w4x2 2wg/cu", 10: "128x128x16 w2x4 2wg/cu", 11: "128x128x32 w4x2",
              12: "128x128x32 w2x4", 13: "128x128x16 w4x2", 14: "t0 ABL1 no-loads", 15: "t0 ABL2 no-loads,no-barrier",
              16: "t0 ABL3 no-barrier", 17: "t6 ABL1 no-loads", 18: "t6 ABL2 no-loads,no-barrier", 19: "t6 ABL3 no-barrier"}
BN = {0: 128, 1: 64, 2: 64, 3: 128, 4: 128, 5: 256, 6: 128, 7: 256, 8: 256}
BN.update({t: 128 for t in range(9, 20)})
TILE_NAMES.update({20: "glds 128x128x32 w2x2", 21: "glds 256x128x16 w4x2", 22: "glds 128x128x16 w4x2",
                   23: "glds 256x128x32 w4x2", 24: "glds 256x256x16 w4x2", 25: "glds 256x256x32 w4x2",
                   26: "glds 128x128x16 w2x2", 27: "glds 256x256x16 w4x4"})
BN.update({20: 128, 21: 128, 22: 128, 23: 128, 24: 256, 25: 256, 26: 128, 27: 256, 28: 64, 29: 64, 30: 64, 31: 128})
TILE_NAMES.update({32: "glds 256x128x16 ABL no-epilogue", 33: "glds 256x128x16 w4x2 3-stage", 34: "glds 128x64x32 w4x2 3-stage",
                   35: "glds 256x128x32 w4x2 3-stage", 36: "glds 64x32x32 w2x1 3-stage", 37: "glds 64x64x32 w2x2 3-stage",
                   38: "glds 128x32x32 w4x1 3-stage", 39: "glds 32x32x32 w1x1 3-stage",
                   40: "glds 256x256x16 w4x4 3-stage", 41: "glds 256x256x16 w4x4 2-stage",
                   42: "t33 with workgroup barriers in the epilogue", 43: "t33 + setprio", 44: "t33 with global_load_lds", 45: "t33 with DMA right behind the barrier",
                   
                   48: "n48 16x16x4 buffer_load..lds", 49: "n48 16x16x4 global_load_lds"})
BN.update({32: 128, 33: 128, 34: 64, 35: 128, 36: 32, 37: 64, 38: 32, 39: 32, 40: 256, 41: 256, 42: 128, 43: 128, 44: 128, 45: 128, 48: 48, 49: 48})
TILE_NAMES.update({28: "glds 128x64x16 w2x2", 29: "glds 128x64x32 w4x2", 30: "glds 128x64x32 w2x2", 31: "glds 128x128x32 w4x2"})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", default="0,3,4,5,6,7,8")
    ap.add_argument("--shapes", default="out,qkv,fc1,fc2,conv3,fc1_nogelu")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--json", default=None)
    ap.add_argument("--bf16", action="store_true", help="sweep the bf16 GEMM instantiations (tile ids 0..6)")
    a = ap.parse_args()
    eng = Engine(seeded_state_dict(0), 0, diag=True)  # libnomad_diag.so: experimental tile ids
    g = torch.Generator().manual_seed(0)
    res = []
    if a.bf16:
        names = {0: "bf16 256x128 w4x2", 1: "bf16 128x128 w4x2", 2: "bf16 128x64 w4x2", 3: "bf16 256x256 w4x2",
                 4: "bf16 64x64 w2x2", 5: "bf16 128x128 w2x2", 6: "bf16 256x128 w2x2",
                 7: "bf16 128x128 ABL no-epilogue", 8: "bf16 128x128 ABL one-k-tile",
                 9: "bf16 256x256 w2x4 bk32 4st", 10: "bf16 256x256 w4x2 bk32 4st", 11: "bf16 128x128 w4x2 bk32 4st",
                 12: "bf16 128x128 w4x2 bk64 3st", 13: "bf16 256x128 w4x2 bk64 3st", 14: "bf16 256x128 w4x2 bk32 4st",
                 15: "bf16 256x256 w2x4 bk64 2st", 16: "bf16 256x256 8-phase", 17: "bf16 256x256 8-phase ABL no-epilogue", 18: "bf16 256x256 8-phase buffer_load..lds", 19: "bf16 256x256 8-phase no setprio", 36: "bf16 256x256 8-phase, timeline probe build",
                 42: "bf16 256x256 8-phase, non-temporal C stores", 43: "bf16 256x256 8-phase, non-temporal C stores + R loads", 44: "bf16 256x256 8-phase, non-temporal R loads", 45: "bf16 256x256 8-phase, probe: A tile 0 for every workgroup", 46: "bf16 256x256 8-phase, 3 B buffers", 47: "bf16 8-phase probe: no LDS-DMA", 48: "bf16 8-phase probe: no DMA, no LDS reads", 49: "bf16 8-phase probe: no LDS reads", 50: "bf16 8-phase probe: no loads, no barriers", 51: "bf16 8-phase, LDS-DMA nt on A", 52: "bf16 8-phase, LDS-DMA nt on B", 53: "bf16 8-phase, LDS-DMA nt on A and B", 54: "bf16 8-phase, LDS-DMA sc1 on A and B",
                 58: "bf16 256x256 8-phase, 3 B buffers, nt stores, residual prefetch, plain epilogue (shipped until round 5)",
                 60: "bf16 256x256 8-phase PERSISTENT, direct epilogue (round 5)", 62: "bf16 persistent, probe: no output stores", 63: "bf16 persistent, every epilogue between tiles (round 5, first step)"}
        bn = {0: 128, 1: 128, 2: 64, 3: 256, 4: 64, 5: 128, 6: 128, 7: 128, 8: 128, 9: 256, 10: 256, 11: 128, 12: 128,
              13: 128, 14: 128, 15: 256, 16: 256, 17: 256, 18: 256, 19: 256, 36: 256, 42: 256, 43: 256, 44: 256, 45: 256, 46: 256, 47: 256, 48: 256, 49: 256, 50: 256, 51: 256, 52: 256, 53: 256, 54: 256, 58: 256, 60: 256, 62: 256, 63: 256, 64: 256, 65: 256, 66: 256, 67: 256}
        for sname in a.shapes.split(","):
            M, N, K, has_b, gelu, has_r = SHAPES[sname]
            A = torch.randn(M, K, generator=g).bfloat16().cuda()
            W = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().cuda()
            b = torch.randn(N, generator=g).cuda() if has_b else None
            R = torch.randn(M, N, generator=g).bfloat16().cuda() if has_r else None
            first = None
            for t in (int(x) for x in a.tiles.split(",")):
                if N % bn[t] or (t in (16, 17, 18, 19, 36, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 58, 60, 62, 63) and K % 128):
                    continue
                out = eng.diag_gemm_bf16(A, W, b, R, gelu=gelu, tile=t).float()
                if first is None:
                    first = out
                diff = (out - first).abs().max().item() / max(first.abs().max().item(), 1e-30)
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.iters + 1)]
                ev[0].record()
                for i in range(a.iters):
                    eng.diag_gemm_bf16(A, W, b, R, gelu=gelu, tile=t)
                    ev[i + 1].record()
                torch.cuda.synchronize()
                ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(a.iters))
                tf = 2.0 * M * N * K / (ms[len(ms) // 2] * 1e-3) / 1e12
                row = {"shape": sname, "tile": t, "cfg": names[t], "ms_med": round(ms[len(ms) // 2], 4),
                       "ms_min": round(ms[0], 4), "tflops": round(tf, 1), "bit_identical": bool(torch.equal(out, first)),
                       "rel_diff_vs_first": diff}
                res.append(row)
                print(json.dumps(row), flush=True)
        if a.json:
            json.dump(res, open(a.json, "w"), indent=1)
        return
    for sname in a.shapes.split(","):
        M, N, K, has_b, gelu, has_r = SHAPES[sname]
        A = torch.randn(M, K, generator=g).cuda()
        W = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
        b = torch.randn(N, generator=g).cuda() if has_b else None
        R = torch.randn(M, N, generator=g).cuda() if has_r else None
        base = None
        for t in (int(x) for x in a.tiles.split(",")):
            if N % BN[t % 100]:
                continue
            out = eng.diag_gemm(A, W, b, R, gelu=gelu, tile=t)  # warm-up
            if base is None:
                base = out.clone()
            ok = bool(torch.equal(out, base))  # same k order in every instantiation -> bit-identical
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.iters + 1)]
            ev[0].record()
            for i in range(a.iters):
                eng.diag_gemm(A, W, b, R, gelu=gelu, tile=t)
                ev[i + 1].record()
            torch.cuda.synchronize()
            ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(a.iters))
            tf = 2.0 * M * N * K / (ms[len(ms) // 2] * 1e-3) / 1e12
            row = {"shape": sname, "tile": t, "cfg": TILE_NAMES[t % 100] + (f" gm{t % 10000 // 100}" if t % 10000 >= 100 else "") + (f" occ{t // 10000}" if t >= 10000 else ""), "ms_med": round(ms[len(ms) // 2], 4),
                   "ms_min": round(ms[0], 4), "tflops": round(tf, 1), "bit_identical": ok}
            res.append(row)
            print(json.dumps(row), flush=True)
        del A, W, b, R
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
