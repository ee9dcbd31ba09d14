#!/usr/bin/env python3
"""Vendor-library yardstick: the hot GEMM shapes of configs C2 (fp32) and C5 (bf16) through torch.matmul (hipBLASLt / rocBLAS
underneath, no epilogue: plain A W^T) next to the hand-written kernels with their fused epilogues, alternating in one process.
Not a product path - it answers "how far from what the vendor's tuned kernels do on THESE shapes is the library".
Usage on the GPU box:  python3 tools/lib_gemm_yardstick.py [--iters 8] > gpurun_out/<tag>/yardstick.json"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from nomad_amd import _lib  # noqa: E402
from nomad_amd.engine import Engine  # noqa: E402
from nomad_amd.weights import seeded_state_dict  # noqa: E402
from gemm_sweep import SHAPES  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=8)
ap.add_argument("--bf16", default="c5_qkv,c5_out,c5_fc1,c5_fc2,c5_conv4_nogelu,c5h_out,c5h_fc2")
ap.add_argument("--f32", default="qkv,out,fc1,fc2,conv3")
args = ap.parse_args()
eng = Engine(seeded_state_dict(0), 0, diag=True)
g = torch.Generator().manual_seed(0)


def timed(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))
    return ts[len(ts) // 2]


out = []
for dtype, names in (("bf16", args.bf16), ("f32", args.f32)):
    for name in [n for n in names.split(",") if n]:
        M, N, K, has_b, gelu, has_r = SHAPES[name]
        A = torch.randn(M, K, generator=g).cuda()
        W = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
        b = torch.randn(N, generator=g).cuda() if has_b else None
        R = torch.randn(M, N, generator=g).cuda() if has_r else None
        if dtype == "bf16":
            A, W = A.bfloat16(), W.bfloat16()
            R = R.bfloat16() if R is not None else None
            fn, tile = eng.lib.nomad_diag_gemm_bf16, 60   # the shipped persistent 256x256 kernel (round 5; 16 / 58 = the one-tile-per-workgroup kernel before it)
        else:
            fn, tile = eng.lib.nomad_diag_gemm, 33         # the shipped 256x128x16 3-stage kernel
        Co = torch.empty(M, N, device="cuda", dtype=A.dtype)  # preallocated: no memset inside the timed region
        ptr = lambda t: t.data_ptr() if t is not None else None
        ours = lambda: _lib.check(fn(eng.ctx, ptr(A), ptr(W), ptr(b), ptr(R), ptr(Co), M, N, K, int(gelu), tile, eng._stream()), "diag_gemm")
        Wt = W.t()
        C = torch.empty(M, N, device="cuda", dtype=A.dtype)
        lib = lambda: torch.matmul(A, Wt, out=C)
        rec = {"shape": name, "dtype": dtype, "M": M, "N": N, "K": K, "epilogue_ours": {"bias": has_b, "gelu": gelu, "residual": has_r}}
        fl = 2.0 * M * N * K
        t_lib, t_ours = [], []
        for _ in range(3):  # alternate
            t_lib.append(timed(lib, args.iters))
            try:
                t_ours.append(timed(ours, args.iters))
            except Exception as e:  # noqa: BLE001
                rec["ours_error"] = repr(e)
                break
        rec["lib_ms"] = min(t_lib)
        rec["lib_tflops"] = fl / min(t_lib) / 1e9
        if t_ours:
            rec["ours_ms"] = min(t_ours)
            rec["ours_tflops"] = fl / min(t_ours) / 1e9
        if dtype == "bf16" and has_r and N % 256 == 0:   # tile 16 with the residual prefetch in the epilogue
            alt = lambda: _lib.check(fn(eng.ctx, ptr(A), ptr(W), ptr(b), ptr(R), ptr(Co), M, N, K, int(gelu), 57, eng._stream()), "diag_gemm")
            t = min(timed(alt, args.iters) for _ in range(3))
            rec["rpre_ms"] = t
            rec["rpre_tflops"] = fl / t / 1e9
        if dtype == "bf16" and N % 192 == 0 and K % 128 == 0:   # the 256 x 192 tiles of the same schedule
            for tname, tid in (("n192", 55), ("n192_two_b", 56)):
                alt = lambda: _lib.check(fn(eng.ctx, ptr(A), ptr(W), ptr(b), ptr(R), ptr(Co), M, N, K, int(gelu), tid, eng._stream()), "diag_gemm")
                t = min(timed(alt, args.iters) for _ in range(3))
                rec[tname + "_ms"] = t
                rec[tname + "_tflops"] = fl / t / 1e9
        out.append(rec)
        print(json.dumps(rec), flush=True)
        del A, W, R, C, Co
        torch.cuda.empty_cache()
