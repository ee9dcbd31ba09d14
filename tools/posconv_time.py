#!/usr/bin/env python3
"""The bf16 forward's positional convolution alone (nomad_diag_posconv_bf16): variant 0 = the grouped GEMM on 128 x 64 tiles, 1 = the
kernel with the input slab resident in LDS (posconv_bf16_slab.hip.h), at configs[4]'s shape (32 x T = 1499) and at 256 x T = 199.
Alternating, events around each launch; TFLOP/s on the algorithmic 2 x M x 768 x 6144 / 16... = 2 M 48 6144 16."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nomad_amd import _lib
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict

eng = Engine(seeded_state_dict(0), 0, diag=True)
lib = eng.lib
_lib.check(lib.nomad_enable_bf16(eng.ctx), "nomad_enable_bf16")
lib.nomad_diag_posconv_bf16.restype = C.c_int
lib.nomad_diag_posconv_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
for B, T in ((32, 1499), (256, 199)):
    gen = torch.Generator().manual_seed(T)
    xpad = torch.zeros(16, B, T + 128, 48)
    xpad[:, :, 64:64 + T] = torch.randn(16, B, T, 48, generator=gen)
    xdev = xpad.bfloat16().cuda()
    y = torch.empty(B * T, 768, dtype=torch.bfloat16, device="cuda")
    res = {"B": B, "T": T}
    st = torch.cuda.current_stream().cuda_stream
    for rep in range(3):
        for v in (0, 1):
            for _ in range(3):
                assert lib.nomad_diag_posconv_bf16(eng.ctx, xdev.data_ptr(), y.data_ptr(), B, T, st, v) == 0
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
            ev[0].record()
            for i in range(20):
                lib.nomad_diag_posconv_bf16(eng.ctx, xdev.data_ptr(), y.data_ptr(), B, T, st, v)
                ev[i + 1].record()
            torch.cuda.synchronize()
            ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(20))
            res.setdefault(f"variant{v}_us_median", []).append(round(ms[10] * 1e3, 1))
            res.setdefault(f"variant{v}_us_min", []).append(round(ms[0] * 1e3, 1))
    fl = 2.0 * B * T * 768 * 6144
    for v in (0, 1):
        res[f"variant{v}_tflops_at_best_median"] = round(fl / (min(res[f"variant{v}_us_median"]) * 1e-6) / 1e12, 1)
    print(json.dumps(res), flush=True)
