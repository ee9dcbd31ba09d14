#!/usr/bin/env python3
"""conv0 + GroupNorm + GELU of the bf16 path alone, 32 x 30 s clips (config C5's launch): the VALU kernel (variant 0) against the
matrix-core kernel at several register budgets / unroll factors (variant 4 = what the forward launches = <4, 1>; 5-7 = <OCC, UF> =
<3,2> <3,1> <3,4>; 9 = timing probe: no GELU)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nomad_amd import _lib
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict
eng = Engine(seeded_state_dict(0), 0, diag=True)
lib = eng.lib
_lib.check(lib.nomad_enable_bf16(eng.ctx), "nomad_enable_bf16")
lib.nomad_diag_conv0_bf16.restype = C.c_int
lib.nomad_diag_conv0_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
B, N = 32, 480000
L0 = (N - 10) // 5 + 1
wav = (0.1 * torch.randn(B, N, generator=torch.Generator().manual_seed(0))).clamp(-1, 1).cuda()
out = torch.empty(B, L0, 512, dtype=torch.bfloat16, device="cuda")
scratch = torch.empty(8 * 65 * B * 16 + 8 * 512 * B + 4096, dtype=torch.uint8, device="cuda")
res = {}
for rep in range(2):
    for v in (0, 4, 5, 6, 7, 9):
        fn = lambda: lib.nomad_diag_conv0_bf16(eng.ctx, wav.data_ptr(), B, N, out.data_ptr(), scratch.data_ptr(), torch.cuda.current_stream().cuda_stream, v)
        for _ in range(3):
            assert fn() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[f"variant{v}_rep{rep}_ms"] = round(e0.elapsed_time(e1) / 20, 4)   # includes wav_stats + gn_fold (~0.05 ms)
print(json.dumps(res))
