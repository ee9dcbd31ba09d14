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


@pytest.mark.parametrize("tile", [37, 20, 31, 33])
def test_x3_product_gemm_repeats_bit_identically_under_its_own_kind_of_load(engine, tile):
    """The bf16x3-products GEMM (gemm_f32_glds_kernel<..., X3>) is itself a 32x32x16-bf16-MFMA kernel with two workgroups per CU
    whose waves spend half their time on VALU work (the hi / lo split: v_cvt_pk_bf16_f32, v_dot2c_f32_bf16) - the very
    co-residency that made v_pk_fma_f32 lose products.  300 launches on a fine-tuning-size problem, with the 128 x 128 bf16
    GEMM running on another stream as well: every result equal to the first, bit for bit."""
    engine.gemm_precision = "bf16x3"
    try:
        g = torch.Generator().manual_seed(tile)
        A = torch.randn(11976, 768, generator=g).cuda()
        W = (torch.randn(768, 768, generator=g) * 0.03).cuda()
        bias = torch.randn(768, generator=g).cuda()
        ref = engine.diag_gemm(A, W, bias=bias, gelu=True, tile=tile).clone()
        Ab, Wb = _aggressor_operands(32, 199)
        side = torch.cuda.Stream()
        bad = 0
        for it in range(300):
            if it % 3 == 0:
                with torch.cuda.stream(side):
                    engine.diag_gemm_bf16(Ab, Wb, tile=1)
            bad += int(not torch.equal(engine.diag_gemm(A, W, bias=bias, gelu=True, tile=tile), ref))
        torch.cuda.synchronize()
        assert bad == 0, f"tile {tile}: {bad} of 300 results differ"
    finally:
        engine.gemm_precision = "fp32"


def test_forward_backward_bf16x3_products_repeat_bit_identically(built_lib, sd0):
    """Nomad(precision="bf16x3").forward() + backward at config C4's shape, 30 times: the loss and d loss / d estimate are the
    same bits every time (two branches on two streams, every GEMM on three bf16 products)."""
    from nomad_amd.nomad import Nomad
    nmd = Nomad(weights=sd0, precision="bf16x3")
    gen = torch.Generator().manual_seed(12)
    clean = (0.1 * torch.randn(32, 1, 16384, generator=gen)).clamp(-1, 1).cuda()
    est0 = (clean + 0.02 * torch.randn(32, 1, 16384, generator=gen).cuda()).clamp(-1, 1)
    first = None
    for it in range(30):
        est = est0.clone().requires_grad_(True)
        loss = nmd.forward(est, clean)
        loss.backward()
        cur = (loss.detach().clone(), est.grad.clone())
        if first is None:
            first = cur
        else:
            assert torch.equal(cur[0], first[0]) and torch.equal(cur[1], first[1]), it
    nmd.engine.close()


def test_forward_backward_fp32_splitk_repeats_bit_identically(built_lib, sd0):
    """Config C4 at the reference's precision (round 4: the no-gradient branch splits K too, on its own per-stream block):
    loss and d loss / d estimate are the same bits on every repeat, the loss under no_grad is the same bits before and
    after a backward has been enabled in the process, and a scoring embed of the same clips never changes."""
    from nomad_amd.nomad import Nomad
    nmd = Nomad(weights=sd0)
    gen = torch.Generator().manual_seed(13)
    clean = (0.1 * torch.randn(32, 1, 16384, generator=gen)).clamp(-1, 1).cuda()
    est0 = (clean + 0.02 * torch.randn(32, 1, 16384, generator=gen).cuda()).clamp(-1, 1)
    emb_before = nmd.model(est0).clone()
    with torch.no_grad():
        loss_nograd_before = nmd.forward(est0, clean).clone()       # both branches on the layer-output forward, two streams
    first = None
    for it in range(12):
        est = est0.clone().requires_grad_(True)
        loss = nmd.forward(est, clean)
        loss.backward()
        cur = (loss.detach().clone(), est.grad.clone())
        if first is None:
            first = cur
        else:
            assert torch.equal(cur[0], first[0]) and torch.equal(cur[1], first[1]), it
    with torch.no_grad():
        assert torch.equal(nmd.forward(est0, clean), loss_nograd_before)
    assert torch.equal(nmd.model(est0), emb_before)
    nmd.engine.close()


def test_pairwise_calls_on_different_streams_do_not_share_scratch(engine):
    """Round-2 advice: nomad_pairwise kept its per-tile row sums in ONE context-owned scratch, so two calls in flight on different
    streams of one context raced.  The scratch is per launch stream now (64 blocks at most, rebound after a drain): 70 streams,
    four calls in flight at a time on different operands, every result bit-equal to its single-stream value."""
    g = torch.Generator().manual_seed(4)
    sets = []
    for k in range(4):
        deg = torch.nn.functional.normalize(torch.randn(900 + 37 * k, 256, generator=g), dim=1).cuda()
        ref = torch.nn.functional.normalize(torch.randn(300 + 11 * k, 256, generator=g), dim=1).cuda()
        d, m = engine.pairwise(deg, ref)
        sets.append((deg, ref, d.clone(), m.clone()))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(70)]
    for base in range(0, 68, 4):
        outs = []
        for k in range(4):
            with torch.cuda.stream(streams[base + k]):
                outs.append(engine.pairwise(sets[k][0], sets[k][1]))
        torch.cuda.synchronize()
        for k in range(4):
            assert torch.equal(outs[k][0], sets[k][2]) and torch.equal(outs[k][1], sets[k][3]), (base, k)


def test_layer_forward_on_three_streams_is_timing_independent(built_lib, sd0):
    """Round-4 advice: whether the small layer-output forward (LossNetLayers, M < 4096 frames) split K used to depend on which of
    two context-wide partial-sum blocks a launch stream could grab (hipEventQuery) - a third stream's forward ran unsplit when the
    other two were still busy, and split-K folds partial sums in another order.  The block is part of each call's own workspace
    now: three such forwards in flight on three streams give the bits of one forward alone, on every repeat."""
    from nomad_amd.engine import Engine
    eng = Engine(sd0, 0)
    gen = torch.Generator().manual_seed(21)
    wavs = [(0.1 * torch.randn(32, 16384, generator=gen)).clamp(-1, 1).cuda() for _ in range(3)]
    alone = []
    for w in wavs:
        e, l = eng.embed(w, want_layers=True)
        alone.append((e.clone(), l.clone()))
    torch.cuda.synchronize()
    streams = [torch.cuda.current_stream(), eng.side_stream(1), eng.side_stream(2)]
    for it in range(6):
        outs = []
        for k in range(3):
            streams[k].wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(streams[k]):
                outs.append(eng.embed(wavs[k], want_layers=True, side=k))
        torch.cuda.synchronize()
        for k in range(3):
            assert torch.equal(outs[k][0], alone[k][0]) and torch.equal(outs[k][1], alone[k][1]), (it, k)
    eng.close()


def test_two_engines_two_host_threads(built_lib, sd0):
    """Round-4 verdict: the tile-choice hint (nomad_set_concurrent_parts) and the CU count were process-wide globals written by
    every context.  They live in the context now: two engines driven from two host threads - one submitting split batches (hint 2),
    one single small batches (hint 1) - never touch each other's state, and each reproduces its single-threaded bits."""
    import threading
    from nomad_amd.engine import Engine
    a, b = Engine(sd0, 0), Engine(sd0, 0)
    gen = torch.Generator().manual_seed(22)
    wa = (0.1 * torch.randn(64, 64000, generator=gen)).clamp(-1, 1).cuda()     # 12 736 frames: Engine.embed splits it on two streams
    wb = (0.1 * torch.randn(3, 16384, generator=gen)).clamp(-1, 1).cuda()
    ra, rb = a.embed(wa).clone(), b.embed(wb).clone()
    torch.cuda.synchronize()
    errs = []

    def work(eng, w, ref, n):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(n):
                    out = eng.embed(w)
                    s.synchronize()
                    if not torch.equal(out, ref):
                        errs.append("mismatch")
        except Exception as e:   # noqa: BLE001
            errs.append(repr(e))

    ta = threading.Thread(target=work, args=(a, wa, ra, 6))
    tb = threading.Thread(target=work, args=(b, wb, rb, 40))
    ta.start(); tb.start(); ta.join(); tb.join()
    assert not errs, errs[:3]
    a.close(); b.close()
