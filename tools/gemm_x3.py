#!/usr/bin/env python3
"""Time the bf16x3 GEMM (split operands, three bf16 MFMA products) on the model's shapes, next to the fp32 and plain
bf16 kernels.  variants (tile id 20 + v in run_gemm_bf16): K-concatenated kernel 0 split out, 1 fp32 out, 2..6 its timing
probes (ABL 3, 4, 5, 6, 1 of gemm_bf16_8phase.hip.h); staged-once kernel 7 split out, 8 fp32 out, 9 no epilogue, 10 no DMA."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict
from gemm_sweep import SHAPES


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="qkv,out,fc1,fc2,conv3")
    ap.add_argument("--variants", default="1,8,1,8,9,10")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    eng = Engine(seeded_state_dict(0), 0, diag=True)  # libnomad_diag.so: experimental tile ids
    g = torch.Generator().manual_seed(0)
    res = []
    for sname in a.shapes.split(","):
        M, N, K, has_b, gelu, has_r = SHAPES[sname]
        A = eng.diag_split_bf16(torch.randn(M, K, device="cuda") if M * K > 5e8 else torch.randn(M, K, generator=g).cuda())
        W = eng.diag_split_bf16((torch.randn(N, K, generator=g) * K ** -0.5).cuda())
        b = torch.randn(N, generator=g).cuda() if has_b else None
        R = eng.diag_split_bf16(torch.randn(M, N, generator=g).cuda()) if has_r else None
        for v in (int(x) for x in a.variants.split(",")):
            out = eng.diag_gemm_bf16x3(A, W, b, R, gelu=gelu, variant=v)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.iters + 1)]
            ev[0].record()
            for i in range(a.iters):
                eng.diag_gemm_bf16x3(A, W, b, R, gelu=gelu, variant=v, out=out)
                ev[i + 1].record()
            torch.cuda.synchronize()
            ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(a.iters))
            med = ms[len(ms) // 2]
            row = {"shape": sname, "M": M, "N": N, "K": K, "variant": v, "ms_med": round(med, 4),
                   "tflops_fp32_equiv": round(2.0 * M * N * K / (med * 1e-3) / 1e12, 1),
                   "tflops_bf16_executed": round(6.0 * M * N * K / (med * 1e-3) / 1e12, 1)}
            res.append(row)
            print(json.dumps(row), flush=True)
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
