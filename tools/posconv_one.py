#!/usr/bin/env python3
"""A few launches of the slab pos-conv at configs[4]'s shape (for rocprofv3 --pmc passes: tools/gpu_pmc_posconv.sh)."""
import ctypes as C, os, sys
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
B, T = 32, 1499
gen = torch.Generator().manual_seed(T)
xpad = torch.zeros(16, B, T + 128, 48)
xpad[:, :, 64:64 + T] = torch.randn(16, B, T, 48, generator=gen)
xdev = xpad.bfloat16().cuda()
y = torch.empty(B * T, 768, dtype=torch.bfloat16, device="cuda")
v = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for _ in range(6):
    assert lib.nomad_diag_posconv_bf16(eng.ctx, xdev.data_ptr(), y.data_ptr(), B, T, torch.cuda.current_stream().cuda_stream, v) == 0
torch.cuda.synchronize()
