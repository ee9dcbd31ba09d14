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


def _oracle_grad(sd, wav, head_w, head_b, G_layers, G_emb):
    w = wav.clone().requires_grad_(True)
    outs = O.lossnet_forward(sd, w, head_w, head_b)
    s = sum((outs[i] * G_layers[i]).sum() for i in range(12)) + (outs[12] * G_emb).sum()
    (grad,) = torch.autograd.grad(s, w)
    return grad


@pytest.mark.parametrize("case", ["emb_only", "layer0_only", "all"])
def test_embed_backward_vs_oracle_autograd(engine, sd0, case):
    """A smooth (linear) functional of the 13 outputs: checks the whole backward chain tightly."""
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
    ref = _oracle_grad(sd0, wav, hw, hb, G_layers, G_emb)
    head = (hw.cuda(), hb.cuda())
    emb, layers, saved = engine.embed_train(wav.cuda(), head)
    emb2, layers2 = engine.embed(wav.cuda(), head=head, want_layers=True)
    assert torch.equal(emb, emb2) and torch.equal(layers, layers2)      # training-mode forward == scoring forward
    dwav = engine.embed_backward(wav.cuda(), layers, saved, G_layers.cuda(), G_emb.cuda(), head).cpu()
    assert torch.isfinite(dwav).all()
    assert _rel(dwav, ref) < 1e-3, _rel(dwav, ref)
    cos = F.cosine_similarity(dwav.flatten(), ref.flatten(), dim=0).item()
    assert cos > 0.999999, cos


def test_forward_is_differentiable_like_the_reference(built_lib, sd0):
    """nomad.forward(estimate, clean).backward(): loss value and estimate.grad vs the HF-autograd golden.
    The L1 terms make the gradient piecewise constant in sign(a - b), so elements whose difference sits at
    the fp32 noise floor flip between implementations (the CPU oracle itself differs from HF by 1.3e-3 of
    the gradient's max): tolerance 5e-3 of max|grad| and cosine > 0.9999."""
    from nomad_amd.nomad import Nomad
    g = np.load(os.path.join(GOLD, "hf_loss.npz"))
    nmd = Nomad(weights=sd0)
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
    assert _rel(grad_nomad, ref) < 5e-3, _rel(grad_nomad, ref)
    assert F.cosine_similarity(grad_nomad.flatten(), ref.flatten(), dim=0).item() > 0.9999
    # without requires_grad the same call still returns the loss value
    assert abs(nmd.forward(est.detach(), clean).item() - float(g["loss"])) < 1e-4
