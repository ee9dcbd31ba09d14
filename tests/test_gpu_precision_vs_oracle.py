"""GPU: the reduced-precision scoring paths (bf16: config C5; bf16x3) against the CPU ORACLE - not against this
library's own fp32 path - at the BASELINE shapes, with tolerances set at <= 3x the error measured on MI355X
(profiles/r02_precision_vs_oracle.txt), plus size-independent properties at config C5's full batch (32 x 480 000)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import nomad_oracle as O

pytestmark = pytest.mark.gpu

#                 embedding max|err|, NOMAD score max|err|      (measured: see the profile file named above)
TOL = {"bf16": (3e-3, 1.2e-3), "bf16x3": (5e-6, 2e-6)}


def _embed(engine, precision, wav):
    return {"bf16": engine.embed_bf16, "bf16x3": engine.embed_bf16x3}[precision](wav)


def _check(engine, sd0, precision, wav, n_deg):
    with torch.no_grad():
        ref = O.triplet_forward(sd0, wav)
    emb = _embed(engine, precision, wav.cuda()).cpu()
    assert torch.isfinite(emb).all() and (emb.norm(dim=1) - 1).abs().max().item() < 1e-5
    e_err = (emb - ref).abs().max().item()
    cos = F.cosine_similarity(emb, ref, dim=1).min().item()
    d, m = engine.pairwise(emb[:n_deg].cuda().contiguous(), emb[n_deg:].cuda().contiguous())
    dref, mref = O.pairwise(ref[:n_deg].numpy(), ref[n_deg:].numpy())
    s_err = max(float(np.abs(d.cpu().numpy() - dref).max()), float(np.abs(m.cpu().numpy() - mref).max()))
    print(f"{precision} vs oracle, {tuple(wav.shape)}: embedding max|err| {e_err:.3e}, min cosine {cos:.7f}, "
          f"score max|err| {s_err:.3e}")
    e_tol, s_tol = TOL[precision]
    assert e_err < e_tol and s_err < s_tol, (e_err, s_err)


@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])
def test_c2_shape_vs_oracle(engine, sd0, precision):
    """Config C2's clip shape (16 kHz x 4 s -> T = 199), 8 clips, 6 deg x 2 ref scores."""
    gen = torch.Generator().manual_seed(0)
    wav = (0.1 * torch.randn(8, 64000, generator=gen)).clamp(-1, 1)
    _check(engine, sd0, precision, wav, 6)


@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])
def test_c5_shape_vs_oracle(engine, sd0, precision):
    """Config C5's clip shape (30 s -> T = 1499: 24 key tiles per attention row), 3 clips, 2 deg x 1 ref scores."""
    gen = torch.Generator().manual_seed(5)
    wav = (0.1 * torch.randn(3, 480000, generator=gen)).clamp(-1, 1)
    _check(engine, sd0, precision, wav, 2)


@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])
def test_c5_full_batch_properties(engine, precision):
    """BASELINE config C5 at full size (32 x 480 000 samples): finite, unit norm, every clip bit-equal to its own
    single-clip call (batch invariance), d(a, a) = 0."""
    gen = torch.Generator().manual_seed(9)
    wav = (0.1 * torch.randn(32, 480000, generator=gen)).clamp(-1, 1).cuda()
    emb = _embed(engine, precision, wav)
    torch.cuda.synchronize()
    assert torch.isfinite(emb).all()
    assert (emb.norm(dim=1) - 1).abs().max().item() < 1e-5
    for i in (0, 17, 31):
        one = _embed(engine, precision, wav[i:i + 1].contiguous())
        assert torch.equal(one[0], emb[i]), i
    d, m = engine.pairwise(emb, emb)
    assert d.diagonal().abs().max().item() == 0.0 and d.max().item() <= 2.0 + 1e-6
    # distinct random clips are far from identical
    off = d + torch.eye(32, device=d.device, dtype=d.dtype) * 10
    assert off.min().item() > 1e-4


def test_c5_one_stream_equals_two_streams_bit_for_bit(engine):
    """Round 6: a batch that runs ALONE (Engine.BF16_SPLIT_ROWS = 0: one stream) takes the persistent GEMM's 192-row tile mode for its
    N = 768 GEMMs (out_proj, fc2: 2.93 instead of 2.2 rounds of tiles), the two-stream default keeps 256-row tiles - the same instantiation,
    the same k order per element: every embedding bit-equal, at configs[4]'s full size."""
    gen = torch.Generator().manual_seed(11)
    wav = (0.1 * torch.randn(32, 480000, generator=gen)).clamp(-1, 1).cuda()
    assert engine.BF16_SPLIT_ROWS
    two = engine.embed_bf16(wav)
    keep = engine.BF16_SPLIT_ROWS
    engine.BF16_SPLIT_ROWS = 0
    try:
        one = engine.embed_bf16(wav)
        one_again = engine.embed_bf16(wav)
    finally:
        engine.BF16_SPLIT_ROWS = keep
    torch.cuda.synchronize()
    assert torch.isfinite(one).all()
    assert torch.equal(one, two) and torch.equal(one, one_again)

