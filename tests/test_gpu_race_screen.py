"""GPU: race screens for the packed-FP32 hazard: with the library built WITH v_pk_fma_f32 & co, a kernel that
uses them loses products in lanes 48-63 while waves of a bf16 128 x 128 MFMA GEMM from another stream share its SIMD - the
round-2 "bf16 nondeterminism" (first seen as embed_bf16 differing run to run with the two-stream batch split on).  The
shipped build has no packed-FP32 instruction; these tests hold that every precision's forward, split ON, gives the same
bits on every call under exactly that load, at the reproducer's shape and at config C5's."""
import ctypes as C

import pytest
import torch

from nomad_amd.weights import num_frames

pytestmark = pytest.mark.gpu


def _aggressor_operands(B, T):
    gen = torch.Generator().manual_seed(5)
    A = (torch.randn(B * T, 768, generator=gen) * 0.5).to(torch.bfloat16).cuda()
    W = (torch.randn(768, 768, generator=gen) * 0.03).to(torch.bfloat16).cuda()
    return A, W


def _load(engine, stream, A, W, junk, launches=4):
    """What tripped the hazard: gemm_bf16_glds_kernel<128,128,4,2> (diag tile 1, in the product library) on another stream,
    plus the uneven rocBLAS load of the older screen."""
    with torch.cuda.stream(stream):
        for _ in range(launches):
            engine.diag_gemm_bf16(A, W, tile=1)
        junk[0] = junk[0] @ junk[0] * 1e-3


@pytest.mark.parametrize("precision", ["bf16", "bf16x3", "fp32"])
def test_forward_split_on_is_bit_identical_under_a_co_running_bf16_gemm(engine, precision):
    gen = torch.Generator().manual_seed(33)
    wav = (0.1 * torch.randn(64, 64000, generator=gen)).clamp(-1, 1).cuda()
    fwd = {"bf16": engine.embed_bf16, "bf16x3": engine.embed_bf16x3, "fp32": engine.embed}[precision]
    assert engine.BF16_SPLIT_ROWS and engine.X3_SPLIT_ROWS and engine.F32_SPLIT_ROWS      # the two-stream split is ON by default
    ref = fwd(wav).clone()
    A, W = _aggressor_operands(32, 199)
    junk = [torch.randn(2048, 2048, device="cuda")]
    side = torch.cuda.Stream()
    for it in range(40):
        _load(engine, side, A, W, junk)
        out = fwd(wav)
        if not torch.equal(out, ref):
            rows = torch.nonzero((out != ref).any(dim=1)).flatten().tolist()
            pytest.fail(f"{precision} iteration {it}: clips {rows} differ from the first result (max|diff| {(out - ref).abs().max().item():.3e})")
    torch.cuda.synchronize()


def test_c5_shape_bf16_split_on_is_bit_identical(engine):
    """BASELINE config C5: 32 clips x 30 s, bf16, the batch as two halves on two streams, background load, 30 calls."""
    gen = torch.Generator().manual_seed(34)
    wav = (0.1 * torch.randn(32, 480000, generator=gen)).clamp(-1, 1).cuda()
    assert engine.BF16_SPLIT_ROWS and 32 * num_frames(480000) >= engine.BF16_SPLIT_ROWS
    ref = engine.embed_bf16(wav).clone()
    A, W = _aggressor_operands(32, 199)
    junk = [torch.randn(2048, 2048, device="cuda")]
    side = torch.cuda.Stream()
    for it in range(30):
        _load(engine, side, A, W, junk, launches=8)
        out = engine.embed_bf16(wav)
        if not torch.equal(out, ref):
            rows = torch.nonzero((out != ref).any(dim=1)).flatten().tolist()
            pytest.fail(f"C5 iteration {it}: clips {rows} differ (max|diff| {(out - ref).abs().max().item():.3e})")
    # each half equals what that half gives on its own, one stream, nothing else running: the split changes no bit
    torch.cuda.synchronize()
    keep = engine.BF16_SPLIT_ROWS
    engine.BF16_SPLIT_ROWS = 0
    try:
        alone = engine.embed_bf16(wav[:16].contiguous())
    finally:
        engine.BF16_SPLIT_ROWS = keep
    assert torch.equal(alone, ref[:16])
    del wav
    engine._ws = None
    engine._ws_side.clear()
    torch.cuda.empty_cache()


def test_conv0_under_the_strongest_reproducer(built_lib, sd0):
    """The victim kernel alone (conv0 + GroupNorm + GELU of the bf16 path) against the aggressors that made it differ in
    97 % of the calls when it was compiled with v_pk_fma_f32 (diag tiles 11 / 12: the 128 x 128 bf16 GEMM with 3 / 4 LDS
    stages; profiles/r03_race_hunt.txt): 0 mismatches in 400 calls."""
    from nomad_amd.engine import Engine
    eng = Engine(sd0, 0, diag=True)
    lib = eng.lib
    lib.nomad_diag_conv0_bf16.restype = C.c_int
    lib.nomad_diag_conv0_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    B, N = 32, 64000
    L0 = (N - 10) // 5 + 1
    gen = torch.Generator().manual_seed(33)
    wav = (0.1 * torch.randn(B, N, generator=gen)).clamp(-1, 1).cuda()
    out = torch.empty(B, L0, 512, dtype=torch.bfloat16, device="cuda")
    scratch = torch.empty(8 * 65 * B * 4 + 8 * 512 * B + 4096, dtype=torch.uint8, device="cuda")

    def conv0():
        assert lib.nomad_diag_conv0_bf16(eng.ctx, wav.data_ptr(), B, N, out.data_ptr(), scratch.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream, 0) == 0
    conv0()
    ref = out.view(torch.int16).clone()
    A, W = _aggressor_operands(B, 199)
    side = torch.cuda.Stream()
    bad = 0
    for it in range(400):
        with torch.cuda.stream(side):
            for _ in range(6):
                eng.diag_gemm_bf16(A, W, tile=11 + (it & 1))
        conv0()
        bad += int(not torch.equal(out.view(torch.int16), ref))
    torch.cuda.synchronize()
    eng.close()
    assert bad == 0, f"conv0 differed from its reference in {bad} of 400 calls"
