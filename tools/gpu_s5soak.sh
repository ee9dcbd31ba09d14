#!/bin/bash
# round 5, last soak: repeated fp32 / bf16 / ragged forwards (tools/soak.py), the persistent bf16 GEMM and the bf16 forwards at 4 s / 30 s
# under aggressor streams (tools/soak_p9.py), the slab pos-conv repeated beside a persistent GEMM on another stream: all bit-identical
TAG=${1:-s5soak}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python3 tools/soak.py 150 > $OUT/soak.txt 2>&1; echo "soak exit $?" | tee -a $OUT/summary.txt; tail -2 $OUT/soak.txt
timeout 1500 python3 tools/soak_p9.py 40 > $OUT/soak_p9.txt 2>&1; echo "soak_p9 exit $?" | tee -a $OUT/summary.txt; tail -4 $OUT/soak_p9.txt
timeout 900 python3 - > $OUT/soak_posconv.txt 2>&1 <<'PY'
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
from nomad_amd import _lib
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict
eng = Engine(seeded_state_dict(0), 0, diag=True)
lib = eng.lib
_lib.check(lib.nomad_enable_bf16(eng.ctx), "nomad_enable_bf16")
lib.nomad_diag_posconv_bf16.restype = C.c_int
lib.nomad_diag_posconv_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
g = torch.Generator().manual_seed(3)
side = torch.cuda.Stream()
A2 = torch.randn(47968, 768, generator=g).bfloat16().cuda()
W2 = (torch.randn(2304, 768, generator=g) * 768 ** -0.5).bfloat16().cuda()
out2 = torch.empty(47968, 2304, dtype=torch.bfloat16, device="cuda")
bad = runs = 0
for B, T in ((32, 1499), (64, 199), (9, 700), (40, 100)):
    xpad = torch.zeros(16, B, T + 128, 48)
    xpad[:, :, 64:64 + T] = torch.randn(16, B, T, 48, generator=g)
    xdev = xpad.bfloat16().cuda()
    y = torch.empty(B * T, 768, dtype=torch.bfloat16, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert lib.nomad_diag_posconv_bf16(eng.ctx, xdev.data_ptr(), y.data_ptr(), B, T, st, 1) == 0
    torch.cuda.synchronize()
    ref = y.clone()
    for i in range(60):
        if i % 2:
            with torch.cuda.stream(side):
                eng.diag_gemm_bf16(A2, W2, None, None, gelu=False, tile=60, out=out2)
        y.fill_(float("nan"))
        assert lib.nomad_diag_posconv_bf16(eng.ctx, xdev.data_ptr(), y.data_ptr(), B, T, st, 1) == 0
        torch.cuda.synchronize()
        runs += 1
        bad += int(not torch.equal(y, ref))
print(f"slab pos-conv: {runs} launches on 4 shapes, every other one beside a persistent GEMM on a second stream: mismatches = {bad}")
sys.exit(1 if bad else 0)
PY
echo "soak_posconv exit $?" | tee -a $OUT/summary.txt; tail -2 $OUT/soak_posconv.txt
