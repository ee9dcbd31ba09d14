"""GPU: d loss / d waveform of Nomad.forward() vs torch.autograd on
the CPU oracle and vs the committed HF-autograd golden."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLD
from oracle import nomad_oracle as O

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


@pytest.mark.parametrize("M,N", [(7, 768), (300, 512)])
def test_layernorm_backward(engine, M, N):
    g = torch.Generator().manual_seed(M)
    x = (torch.randn(M, N, generator=g) * 2 + 0.3).double().requires_grad_(True)
    gamma, beta = 1 + 0.1 * torch.randn(N, generator=g), 0.1 * torch.randn(N, generator=g)
    up = torch.randn(M, N, generator=g)
    y = F.layer_norm(x, (N,), gamma.double(), beta.double(), 1e-5)
    (ref,) = torch.autograd.grad((y * up.double()).sum(), x)
    dx = engine.diag_layernorm_bwd(x.detach().float().cuda(), up.cuda(), gamma.cuda()).cpu()
    assert _rel(dx.double(), ref) < 2e-5


@pytest.mark.parametrize("B,T", [(2, 18), (1, 50), (1, 70)])
def test_attention_backward(engine, B, T):
    g = torch.Generator().manual_seed(T)
    qkv = (torch.randn(B * T, 2304, generator=g) * 0.5).double().requires_grad_(True)
    dctx = torch.randn(B * T, 768, generator=g)
    q, k, v = (qkv[:, i * 768:(i + 1) * 768].view(B, T, 12, 64).transpose(1, 2) for i in range(3))
    ctx = (torch.softmax(q @ k.transpose(-1, -2), -1) @ v).transpose(1, 2).reshape(B * T, 768)
    (ref,) = torch.autograd.grad((ctx * dctx.double()).sum(), qkv)
    out, lse, dqkv = engine.diag_attention_bwd(qkv.detach().float().cuda(), dctx.cuda(), B, T)
    assert (out.cpu().double() - ctx.detach()).abs().max().item() < 1e-5
    lse_ref = torch.logsumexp(q @ k.transpose(-1, -2), -1).reshape(B * 12, T)
    assert (lse.cpu().double() - lse_ref.detach()).abs().max().item() < 1e-5
    assert _rel(dqkv.cpu().double(), ref) < 5e-5


@pytest.mark.parametrize("B,T", [(3, 50), (1, 64), (2, 1), (1, 17), (2, 33)])
def test_fused_attention_backward_of_short_clips_is_bit_identical(built_lib, sd0, monkeypatch, B, T):
    """Round 6: clips of at most 64 frames take ONE fused launch (attn_bwd_small_kernel) instead of rowdot + dkv + dq; every product and
    sum is the three-kernel path's, in its order - the same bits (NOMAD_ATTN_BWD_SMALL=0 on the diag library = the three kernels)."""
    from nomad_amd.engine import Engine
    g = torch.Generator().manual_seed(100 + T)
    qkv = (torch.randn(B * T, 2304, generator=g) * 0.5).cuda()
    dctx = torch.randn(B * T, 768, generator=g).cuda()
    outs = []
    for flag in ("0", "1"):
        monkeypatch.setenv("NOMAD_ATTN_BWD_SMALL", flag)
        eng = Engine(sd0, 0, diag=True)
        out, lse, dqkv = eng.diag_attention_bwd(qkv, dctx, B, T)
        torch.cuda.synchronize()
        outs.append((out.clone(), lse.clone(), dqkv.clone()))
        eng.close()
    assert torch.isfinite(outs[1][2]).all()
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


def _oracle_grad(sd, wav, head_w, head_b, G_layers, G_emb, mult=1.0):
    w = wav.clone().requires_grad_(True)
    outs = O.lossnet_forward(sd, w, head_w, head_b, feature_grad_mult=mult, required_seq_len_multiple=2)
    s = sum((outs[i] * G_layers[i]).sum() for i in range(12)) + (outs[12] * G_emb).sum()
    (grad,) = torch.autograd.grad(s, w)
    return grad


@pytest.fixture
def grad_mult(engine):
    """Sets fairseq's feature_grad_mult on the shared engine for one test and restores the default (0.1) afterwards."""
    default = engine.feature_grad_mult
    assert abs(default - 0.1) < 1e-8          # wav2vec 2.0 BASE / wav2vec_small.pt

    def set_(m):
        engine.feature_grad_mult = m
    yield set_
    engine.feature_grad_mult = default


@pytest.mark.parametrize("mult", [1.0, 0.1])
@pytest.mark.parametrize("case", ["emb_only", "layer0_only", "all"])
def test_embed_backward_vs_oracle_autograd(engine, sd0, case, mult, grad_mult):
    """A smooth (linear) functional of the 13 outputs: checks the whole backward chain tightly, with the plain chain
    rule (1.0) and with the reference model's GradMultiply(features, 0.1)."""
    grad_mult(mult)
    gen = torch.Generator().manual_seed(11)
    B, N = 2, 6000
    wav = (0.1 * torch.randn(B, N, generator=gen)).clamp(-1, 1)
    T = 18
    hw = (torch.rand(256, 768, generator=gen) * 2 - 1) / 768 ** 0.5
    hb = (torch.rand(256, generator=gen) * 2 - 1) / 768 ** 0.5
    G_layers = torch.randn(12, B, T, 768, generator=gen) / (B * T * 768)
    G_emb = torch.randn(B, 256, generator=gen) / (B * 256)
    if case == "emb_only":
        G_layers.zero_()
    elif case == "layer0_only":
        G_layers[1:].zero_()
        G_emb.zero_()
    ref = _oracle_grad(sd0, wav, hw, hb, G_layers, G_emb, mult)
    head = (hw.cuda(), hb.cuda())
    emb, layers, saved = engine.embed_train(wav.cuda(), head)
    emb2, layers2 = engine.embed(wav.cuda(), head=head, want_layers=True)
    # the loss forward may split the contraction of its small-M GEMMs (a different summation order than the scoring
    # forward, whose bits must not depend on the batch): equal to rounding, not bit for bit
    assert (emb - emb2).abs().max().item() < 1e-6 and (layers - layers2).abs().max().item() < 2e-5
    dwav = engine.embed_backward(wav.cuda(), layers, saved, G_layers.cuda(), G_emb.cuda(), head).cpu()
    assert torch.isfinite(dwav).all()
    assert _rel(dwav, ref) < 1e-3, _rel(dwav, ref)
    cos = F.cosine_similarity(dwav.flatten(), ref.flatten(), dim=0).item()
    assert cos > 0.999999, cos


def test_feature_grad_mult_knob(engine, sd0, grad_mult):
    """The knob is a pure scale of the extractor gradient: grad(0.1) = 0.1 * grad(1.0) to rounding, 0 gives no
    gradient (fairseq: extractor under no_grad), negative values are rejected, forward values never change."""
    gen = torch.Generator().manual_seed(5)
    wav = (0.1 * torch.randn(2, 6320, generator=gen)).clamp(-1, 1).cuda()
    G_layers = (torch.randn(12, 2, 19, 768, generator=gen) / (2 * 19 * 768)).cuda()
    G_emb = (torch.randn(2, 256, generator=gen) / 512).cuda()
    grads, embs = {}, {}
    for m in (1.0, 0.1, 0.0):
        grad_mult(m)
        assert abs(engine.feature_grad_mult - m) < 1e-8
        emb, layers, saved = engine.embed_train(wav)
        embs[m] = emb
        grads[m] = engine.embed_backward(wav, layers, saved, G_layers, G_emb)
    assert torch.equal(embs[1.0], embs[0.1]) and torch.equal(embs[1.0], embs[0.0])
    assert _rel(grads[0.1], 0.1 * grads[1.0]) < 2e-6
    assert grads[1.0].abs().max().item() > 0 and torch.count_nonzero(grads[0.0]).item() == 0
    with pytest.raises(Exception):
        engine.feature_grad_mult = -1.0


@pytest.mark.parametrize("mult,key", [(0.1, "long_grad_fgm01"), (1.0, "long_grad_fgm1")])
def test_full_chain_backward_over_three_attention_tiles(engine, sd0, mult, key, grad_mult):
    """T = 131 frames (odd, three 64-key attention tiles, several split-K chunks): the whole dX chain against the CPU
    oracle's autograd AND against the committed HF-autograd golden with fairseq's GradMultiply hooked onto HF's feature
    extractor (oracle/make_golden.py fgm)."""
    from oracle.make_golden import smooth_functional_weights
    grad_mult(mult)
    g = np.load(os.path.join(GOLD, "hf_grad_fgm.npz"))
    gl = np.load(os.path.join(GOLD, "hf_loss.npz"))
    wav = torch.from_numpy(g["long_wav"])
    T = 131
    G_layers, G_emb = smooth_functional_weights(1, T, int(g["g_seed"]))
    hw, hb = torch.from_numpy(gl["emb_w"]), torch.from_numpy(gl["emb_b"])
    head = (hw.cuda(), hb.cuda())
    emb, layers, saved = engine.embed_train(wav.cuda(), head)
    assert layers.shape == (12, 1, T, 768)
    dwav = engine.embed_backward(wav.cuda(), layers, saved, G_layers.cuda(), G_emb.cuda(), head).cpu()
    ref_oracle = _oracle_grad(sd0, wav, hw, hb, G_layers, G_emb, mult)
    ref_hf = torch.from_numpy(g[key])
    assert _rel(dwav, ref_oracle) < 1e-3, _rel(dwav, ref_oracle)
    assert _rel(dwav, ref_hf) < 1e-3, _rel(dwav, ref_hf)
    assert F.cosine_similarity(dwav.flatten(), ref_hf.flatten(), dim=0).item() > 0.999999


def test_forward_is_differentiable_like_the_reference(built_lib, sd0):
    """nomad.forward(estimate, clean).backward(): loss value and estimate.grad vs the HF-autograd golden, with the
    plain chain rule (feature_grad_mult = 1: what HF autograd computes) and with the reference model's 0.1 (the
    default; golden from HF + a GradMultiply hook).
    The L1 terms make the gradient piecewise constant in sign(a - b), so elements whose difference sits at
    the fp32 noise floor flip between implementations (the CPU oracle itself differs from HF by 1.3e-3 of
    the gradient's max): tolerance 3e-3 of max|grad| and cosine > 0.9999."""
    from nomad_amd.nomad import Nomad
    g = np.load(os.path.join(GOLD, "hf_loss.npz"))
    gf = np.load(os.path.join(GOLD, "hf_grad_fgm.npz"))
    nmd01 = Nomad(weights=sd0)                      # default: the reference model's feature_grad_mult
    assert abs(nmd01.engine.feature_grad_mult - 0.1) < 1e-8
    nmd01.lossnet_layers.embedding_weight = torch.from_numpy(g["emb_w"]).cuda()
    nmd01.lossnet_layers.embedding_bias = torch.from_numpy(g["emb_b"]).cuda()
    est = torch.from_numpy(g["estimate"]).cuda().requires_grad_(True)
    nmd01.forward(est, torch.from_numpy(g["clean"]).cuda()).backward()
    ref01 = torch.from_numpy(gf["grad_l1_fgm01"]).cuda()
    print("L1 gradient at feature_grad_mult 0.1: rel err vs HF+GradMultiply golden", _rel(est.grad, ref01))
    assert _rel(est.grad, ref01) < 3e-3, _rel(est.grad, ref01)
    assert F.cosine_similarity(est.grad.flatten(), ref01.flatten(), dim=0).item() > 0.9999
    del nmd01
    nmd = Nomad(weights=sd0, feature_grad_mult=1.0)
    nmd.lossnet_layers.embedding_weight = torch.from_numpy(g["emb_w"]).cuda()
    nmd.lossnet_layers.embedding_bias = torch.from_numpy(g["emb_b"]).cuda()
    est = torch.from_numpy(g["estimate"]).cuda().requires_grad_(True)
    clean = torch.from_numpy(g["clean"]).cuda()
    mse = F.mse_loss(est, clean)
    loss = mse + 0.5 * nmd.forward(est, clean)          # used as an auxiliary loss, as in nomad_loss_test.py:69
    loss.backward()
    assert abs((loss - mse).item() / 0.5 - float(g["loss"])) < 1e-4
    grad_nomad = (est.grad - 2 * (est - clean).detach() / est.numel()) / 0.5
    ref = torch.from_numpy(g["grad"]).cuda()
    assert grad_nomad.shape == ref.shape == (2, 1, 16384)
    print("L1 gradient at feature_grad_mult 1.0: rel err vs HF golden", _rel(grad_nomad, ref))
    assert _rel(grad_nomad, ref) < 3e-3, _rel(grad_nomad, ref)
    assert F.cosine_similarity(grad_nomad.flatten(), ref.flatten(), dim=0).item() > 0.9999
    # without requires_grad the same call still returns the loss value
    assert abs(nmd.forward(est.detach(), clean).item() - float(g["loss"])) < 1e-4


def test_forward_gradient_with_respect_to_clean(built_lib, sd0):
    """The reference's forward (nomad.py:142-146) is differentiable in BOTH arguments.  |e - c| is symmetric, so d forward(e, c) / d c must equal
    d forward(c, e) / d (first argument) - the path the HF goldens above pin; with both inputs requiring a gradient, both come back."""
    from nomad_amd.nomad import Nomad
    g = np.load(os.path.join(GOLD, "hf_loss.npz"))
    nmd = Nomad(weights=sd0)
    nmd.lossnet_layers.embedding_weight = torch.from_numpy(g["emb_w"]).cuda()
    nmd.lossnet_layers.embedding_bias = torch.from_numpy(g["emb_b"]).cuda()
    est0, cln0 = torch.from_numpy(g["estimate"]).cuda(), torch.from_numpy(g["clean"]).cuda()
    # reference: the clean waveform in the differentiated (first) slot
    a = cln0.clone().requires_grad_(True)
    nmd.forward(a, est0).backward()
    ref_c = a.grad.clone()
    e = est0.clone().requires_grad_(True)
    nmd.forward(e, cln0).backward()
    ref_e = e.grad.clone()
    # clean only
    c = cln0.clone().requires_grad_(True)
    loss = nmd.forward(est0, c)
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    loss.backward()
    assert c.grad is not None and c.grad.shape == cln0.shape
    assert _rel(c.grad, ref_c) < 3e-3, _rel(c.grad, ref_c)
    assert F.cosine_similarity(c.grad.flatten(), ref_c.flatten(), dim=0).item() > 0.9999
    # both
    e2, c2 = est0.clone().requires_grad_(True), cln0.clone().requires_grad_(True)
    nmd.forward(e2, c2).backward()
    assert _rel(e2.grad, ref_e) < 3e-3 and _rel(c2.grad, ref_c) < 3e-3, (_rel(e2.grad, ref_e), _rel(c2.grad, ref_c))


def test_graphed_loss_replays_bit_identically(built_lib, sd0):
    """Round 5 (config C4): nomad.forward + backward captured once as a HIP graph (Nomad.graphed_loss) and replayed on new inputs
    gives the bits of the eager call - same kernels in the same order; every entry point on the path is capture-safe (the split-K
    block of the layer forward used to be bound to a stream by hipEventQuery, which a capture cannot contain)."""
    import torch
    from nomad_amd.nomad import Nomad
    nmd = Nomad(weights=sd0)
    gen = torch.Generator().manual_seed(31)
    shape = (8, 1, 16384)
    clean = [(0.1 * torch.randn(*shape, generator=gen)).clamp(-1, 1).cuda() for _ in range(3)]
    est = [(c + 0.02 * torch.randn(*shape, generator=gen).cuda()).clamp(-1, 1) for c in clean]
    eager = []
    for e0, c in zip(est, clean):
        e = e0.clone().requires_grad_(True)
        loss = nmd.forward(e, c)
        loss.backward()
        eager.append((loss.detach().clone(), e.grad.clone()))
    graphed = nmd.graphed_loss(est[0], clean[0])
    for rep in range(2):
        for k in (2, 0, 1):
            loss, grad = graphed.step(est[k], clean[k])
            torch.cuda.synchronize()
            assert torch.equal(loss, eager[k][0]) and torch.equal(grad, eager[k][1]), (rep, k)
    # ... and as an autograd node inside a larger graph: d (2 loss) / d estimate = 2 * grad
    e = est[1].clone().requires_grad_(True)
    (2.0 * graphed(e, clean[1])).backward()
    assert torch.equal(e.grad, 2.0 * eager[1][1])
    # the restrictions of a captured graph are checked, not just documented: a replaced head tensor (the graph holds the
    # old address) and dropout / LayerDrop (the graph would replay one set of masks) are refused
    keep = nmd.lossnet_layers.embedding_weight
    nmd.lossnet_layers.embedding_weight = keep.clone()
    with pytest.raises(RuntimeError, match="replaced after capture"):
        graphed.step(est[0], clean[0])
    nmd.lossnet_layers.embedding_weight = keep
    graphed.step(est[0], clean[0])
    nmd.engine.enable_backward()
    nmd.engine.train_set_stochastic(dropout=0.1, seed=3)
    with pytest.raises(RuntimeError, match="dropout / LayerDrop"):
        graphed.step(est[0], clean[0])
    with pytest.raises(RuntimeError, match="dropout / LayerDrop"):
        nmd.graphed_loss(est[0], clean[0])
    nmd.engine.train_set_stochastic()
    loss, grad = graphed.step(est[0], clean[0])
    torch.cuda.synchronize()
    assert torch.equal(loss, eager[0][0]) and torch.equal(grad, eager[0][1])
    nmd.engine.close()


def test_splitk_epilogue_with_its_layernorm_is_bit_identical(built_lib, sd0, monkeypatch):
    """configs[3]'s small-M residual GEMMs split K; their epilogue (slices + bias + residual) now also normalises the row
    (splitk_epilogue_ln_kernel: the stand-alone epilogue's sums, the stand-alone LayerNorm's row code).  Against the two-launch form
    (NOMAD_SPLITK_LN=0 on the diag library): the 12 layer outputs, the embedding and d loss / d waveform bit for bit."""
    from nomad_amd.engine import Engine
    gen = torch.Generator().manual_seed(5)
    wav = (0.1 * torch.randn(8, 16384, generator=gen)).clamp(-1, 1).cuda()
    outs = []
    for flag in ("0", "1"):
        monkeypatch.setenv("NOMAD_SPLITK_LN", flag)
        eng = Engine(sd0, 0, diag=True)
        emb, layers, saved = eng.embed_train(wav)
        dl = torch.ones_like(layers) / layers.numel()
        de = torch.ones_like(emb) / emb.numel()
        dwav = eng.embed_backward(wav, layers, saved, dl, de)
        torch.cuda.synchronize()
        outs.append((emb.clone(), layers.clone(), dwav.clone()))
        eng.close()
    assert torch.isfinite(outs[0][1]).all()
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


def test_layernorm_backward_inside_the_split_k_epilogue_is_bit_identical(built_lib, sd0, monkeypatch):
    """Round 6: in the dX-only backward of a small batch the GEMMs in front of a LayerNorm backward (fc1^T, and qkv^T in front of the layer
    below's LN2 / the encoder LN) split K; the LayerNorm-backward kernel then forms their output row from the partial products itself
    (layernorm_bwd_kernel<3, true>) - 25 epilogue launches fewer per configs[3] step.  Against NOMAD_SPLITK_LNB=0 (epilogue + stand-alone
    kernel): d loss / d waveform bit for bit, with and without layer-output gradients, and for a batch too large to split (unchanged path)."""
    from nomad_amd.engine import Engine
    gen = torch.Generator().manual_seed(6)
    for B, n, with_layers in ((8, 16384, True), (3, 9000, False), (40, 64000, True)):
        wav = (0.1 * torch.randn(B, n, generator=gen)).clamp(-1, 1).cuda()
        outs = []
        for flag in ("0", "1"):
            monkeypatch.setenv("NOMAD_SPLITK_LNB", flag)
            eng = Engine(sd0, 0, diag=True)
            emb, layers, saved = eng.embed_train(wav)
            g2 = torch.Generator().manual_seed(7)
            dl = (torch.randn(layers.shape, generator=g2) / layers.numel()).cuda() if with_layers else None
            de = (torch.randn(emb.shape, generator=g2) / emb.numel()).cuda()
            dwav = eng.embed_backward(wav, layers, saved, dl, de)
            torch.cuda.synchronize()
            outs.append(dwav.clone())
            eng.close()
        assert torch.isfinite(outs[0]).all() and outs[0].abs().max() > 0
        assert torch.equal(outs[0], outs[1]), (B, n)

