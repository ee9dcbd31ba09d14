"""GPU: the C ABI's error contract (include/nomad_hip.h: 0 / negative status + nomad_last_error, nothing thrown, nothing
launched on a rejected call, the context usable afterwards) and the smallest inputs the path accepts.

The reference's counterpart of these cases: a clip shorter than the conv stack's receptive field (400 samples) makes
fairseq's feature extractor raise ("Kernel size can't be greater than actual input size", the oracle raises the same way);
the shortest legal clips (T = 1, 2, 3 frames) go through `TripletModel.forward` (`nomad.py:224-231`) like any other."""
import ctypes as C

import pytest
import torch

from nomad_amd import _lib
from nomad_amd._lib import NomadHipError
from oracle import nomad_oracle as O

pytestmark = pytest.mark.gpu


def _wav(B, N, seed=5):
    g = torch.Generator().manual_seed(seed)
    return (0.1 * torch.randn(B, N, generator=g)).clamp(-1, 1)


@pytest.mark.parametrize("N", [400, 719, 720, 1040])
def test_shortest_clips_vs_oracle(engine, sd0, N):
    """T = 1 (one frame: attention over a single key, the pos-conv sees only its padding), 2 and 3 frames."""
    wav = _wav(3, N)
    ref = O.triplet_forward(sd0, wav)
    emb = engine.embed(wav.cuda()).cpu()
    assert (emb - ref).abs().max().item() < 1e-5
    rag = engine.embed_ragged([w for w in wav.cuda()]).cpu()          # the ragged entry point on the same clips
    assert torch.equal(rag, emb)
    x3 = engine.embed_bf16x3(wav.cuda()).cpu()
    assert (x3 - ref).abs().max().item() < 5e-5
    b16 = engine.embed_bf16(wav.cuda()).cpu()
    assert (b16 - ref).abs().max().item() < 5e-3


def test_too_short_clip_is_rejected_everywhere(engine, sd0):
    """399 samples: no frame survives the seven convolutions.  Host wrappers say so before calling; the C entry points
    return NOMAD_ERR_INVALID themselves (a C caller has no wrapper)."""
    wav = _wav(2, 399).cuda()
    with pytest.raises(RuntimeError):
        O.triplet_forward(sd0, wav.cpu())               # what the reference side does with it
    for fn in (engine.embed, engine.embed_bf16, engine.embed_bf16x3):
        with pytest.raises(ValueError, match="shorter than"):
            fn(wav)
    lib, n = engine.lib, C.c_size_t(0)
    for name in ("nomad_workspace_bytes", "nomad_workspace_bytes_bf16", "nomad_workspace_bytes_bf16x3"):
        rc = getattr(lib, name)(engine.ctx, 2, 399, C.byref(n))
        assert rc == _lib.NOMAD_ERR_INVALID and b"bad shape" in lib.nomad_last_error()
    emb = torch.empty(2, 256, device="cuda")
    ws = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    rc = lib.nomad_embed(engine.ctx, wav.data_ptr(), 2, 399, None, None, emb.data_ptr(), None, ws.data_ptr(), ws.numel(), None)
    assert rc == _lib.NOMAD_ERR_INVALID and b"n_samples=399" in lib.nomad_last_error()


def test_bad_arguments_return_status_and_leave_the_context_usable(engine):
    lib = engine.lib
    wav = _wav(2, 4000).cuda()
    good = engine.embed(wav).clone()
    emb = torch.full((2, 256), 7.0, device="cuda")
    need = engine.workspace_bytes(2, 4000)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    args = lambda B=2, w=wav.data_ptr(), e=emb.data_ptr(), wsp=ws.data_ptr(), nb=need: (
        engine.ctx, w, B, 4000, None, None, e, None, wsp, nb, None)
    assert lib.nomad_embed(*args(B=0)) == _lib.NOMAD_ERR_INVALID                       # empty batch
    assert lib.nomad_embed(*args(w=None)) == _lib.NOMAD_ERR_INVALID                    # null input
    assert lib.nomad_embed(*args(e=None)) == _lib.NOMAD_ERR_INVALID                    # null output
    assert lib.nomad_embed(*args(wsp=None)) == _lib.NOMAD_ERR_INVALID                  # null workspace
    rc = lib.nomad_embed(*args(nb=need - 1))                                           # workspace one byte short
    assert rc == _lib.NOMAD_ERR_WORKSPACE and b"workspace" in lib.nomad_last_error()
    torch.cuda.synchronize()
    assert bool((emb == 7.0).all())                                                    # nothing was launched
    mean = torch.empty(4, dtype=torch.float64, device="cuda")
    e4 = torch.randn(4, 256, device="cuda")
    assert lib.nomad_pairwise(engine.ctx, e4.data_ptr(), 0, e4.data_ptr(), 4, None, mean.data_ptr(), None) == _lib.NOMAD_ERR_INVALID
    assert lib.nomad_pairwise(engine.ctx, e4.data_ptr(), 4, e4.data_ptr(), 0, None, mean.data_ptr(), None) == _lib.NOMAD_ERR_INVALID
    assert b"Nr=0" in lib.nomad_last_error()
    assert lib.nomad_pairwise(engine.ctx, e4.data_ptr(), 4, e4.data_ptr(), 4, None, None, None) == _lib.NOMAD_ERR_INVALID
    with pytest.raises(NomadHipError, match="nomad_pairwise"):
        _lib.check(lib.nomad_pairwise(engine.ctx, None, 4, e4.data_ptr(), 4, None, mean.data_ptr(), None), "nomad_pairwise")
    # after all of that the same context still computes the same bits
    assert lib.nomad_embed(*args()) == 0
    torch.cuda.synchronize()
    assert torch.equal(emb, good)


def test_ragged_rejects_a_clip_below_the_receptive_field(engine):
    clips = [c for c in _wav(2, 4000).cuda()] + [_wav(1, 399)[0].cuda()]
    with pytest.raises((ValueError, NomadHipError)):
        engine.embed_ragged(clips)
    out = engine.embed_ragged(clips[:2])            # and the context is fine afterwards
    assert torch.equal(out, engine.embed(torch.stack(clips[:2])))
