#!/usr/bin/env python3
"""Round 5: the persistent bf16 GEMM (gemm_bf16_p9.hip.h, tile 60) against the one-tile-per-workgroup kernel it replaces (tile 58)
and hipBLASLt (torch.matmul, no epilogue) on config C5's shapes - output buffers reused, events around the launches only,
alternating, median of `--iters` launches per round.  Also checks bit-identity of 60 vs 58 on random data.
Usage: python tools/p9_ab.py [--shapes c5_qkv,...] [--tiles 58,60] [--rounds 3]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict
from gemm_sweep import SHAPES


def timed(fn, iters):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    fn()
    torch.cuda.synchronize()
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))
    return ms[len(ms) // 2], ms[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="c5_qkv,c5_out,c5_fc1,c5_fc2,c5_conv4,c5_fc1_nogelu,c5h_out,c5h_fc2,c5_k128")
    ap.add_argument("--tiles", default="58,60")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--vendor", type=int, default=1)
    a = ap.parse_args()
    eng = Engine(seeded_state_dict(0), 0, diag=True)
    g = torch.Generator().manual_seed(0)
    tiles = [int(t) for t in a.tiles.split(",")]
    for sname in a.shapes.split(","):
        M, N, K, has_b, gelu, has_r = SHAPES[sname]
        A = torch.randn(M, K, generator=g).bfloat16().cuda()
        W = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().cuda()
        b = torch.randn(N, generator=g).cuda() if has_b else None
        R = torch.randn(M, N, generator=g).bfloat16().cuda() if has_r else None
        outs = {t: torch.zeros(M, N, dtype=torch.bfloat16, device="cuda") for t in tiles}
        for t in tiles:
            eng.diag_gemm_bf16(A, W, b, R, gelu=gelu, tile=t, out=outs[t])
        torch.cuda.synchronize()
        same = {t: bool(torch.equal(outs[t], outs[tiles[0]])) for t in tiles}
        ref = (A.float() @ W.float().t())
        if b is not None:
            ref += b
        if gelu:
            ref = torch.nn.functional.gelu(ref)
        if R is not None:
            ref += R.float()
        err = {t: float((outs[t].float() - ref).abs().max()) for t in tiles}
        del ref
        best = {t: 1e9 for t in tiles}
        vend = 1e9
        WT = W.t().contiguous() if a.vendor else None
        vout = torch.empty(M, N, dtype=torch.bfloat16, device="cuda") if a.vendor else None
        for _ in range(a.rounds):
            for t in tiles:
                med, _ = timed(lambda: eng.diag_gemm_bf16(A, W, b, R, gelu=gelu, tile=t, out=outs[t]), a.iters)
                best[t] = min(best[t], med)
            if a.vendor:
                med, _ = timed(lambda: torch.matmul(A, WT, out=vout), a.iters)
                vend = min(vend, med)
        fl = 2.0 * M * N * K
        row = {"shape": sname, "M": M, "N": N, "K": K, "epilogue": ("bias" if has_b else "") + ("+gelu" if gelu else "") + ("+residual" if has_r else "")}
        for t in tiles:
            row[f"tile{t}_us"] = round(best[t] * 1e3, 1)
            row[f"tile{t}_tf"] = round(fl / best[t] / 1e9, 1)
            row[f"tile{t}_bit_identical_to_{tiles[0]}"] = same[t]
            row[f"tile{t}_max_abs_err_vs_fp32"] = round(err[t], 5)
        if a.vendor:
            row["vendor_us"] = round(vend * 1e3, 1)
            row["vendor_tf"] = round(fl / vend / 1e9, 1)
        print(json.dumps(row), flush=True)
        del A, W, b, R, outs, WT, vout
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
