"""GPU: the "bf16x3 products on fp32 buffers" mode of the fp32-layout GEMMs (nomad_set_gemm_precision, gemm_f32_glds_kernel
<..., X3>): each product as a_hi w_hi + a_hi w_lo + a_lo w_hi on v_mfma_f32_32x32x16_bf16, hi / lo split in registers.
Kernel level vs float64 (exact on small integers, ~3e-5 relative on random data, every epilogue feature), tile choice
changes no bit; path level: Nomad(precision="bf16x3").forward() - loss and d loss / d estimate - against the fp32 path,
the CPU oracle's autograd and the HF goldens, at the tolerances of tests/test_gpu_backward.py."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLD
from oracle import nomad_oracle as O

pytestmark = pytest.mark.gpu
TILES = [37, 20, 31, 34, 33]      # the instantiations pick_tile() can select (64x64, 128x128 4/8 waves, 128x64, 256x128)


@pytest.fixture
def x3(engine):
    engine.gemm_precision = "bf16x3"
    yield engine
    engine.gemm_precision = "fp32"


def test_knob_roundtrip(engine):
    assert engine.gemm_precision == "fp32"
    engine.gemm_precision = "bf16x3"
    assert engine.gemm_precision == "bf16x3"
    engine.gemm_precision = "fp32"
    with pytest.raises(ValueError):
        engine.gemm_precision = "fp16"


@pytest.mark.parametrize("tile", TILES)
def test_exact_on_small_integers_with_asymmetric_operands(x3, tile):
    """|values| <= 127 are exact in bf16 (lo = 0), products and sums exact in fp32: any fragment / k-order / MFMA layout
    mistake shows as a wrong integer."""
    g = torch.Generator().manual_seed(tile)
    M, N, K = 300, 256, 96
    A = torch.randint(-127, 128, (M, K), generator=g).float()
    W = torch.randint(-5, 6, (N, K), generator=g).float()
    W[:, ::7] *= 3
    out = x3.diag_gemm(A.cuda(), W.cuda(), tile=tile).cpu()
    assert torch.equal(out, A @ W.t())


@pytest.mark.parametrize("tile", TILES)
def test_random_data_vs_float64_with_every_epilogue(x3, tile):
    g = torch.Generator().manual_seed(100 + tile)
    M, N, K = 777, 384, 768
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.05
    bias, R = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ref = F.gelu(A.double() @ W.double().t() + bias.double()) + R.double()
    out = x3.diag_gemm(A.cuda(), W.cuda(), bias=bias.cuda(), R=R.cuda(), gelu=True, tile=tile).cpu().double()
    plain = x3.diag_gemm(A.cuda(), W.cuda(), tile=tile).cpu().double()
    scale = (A.double() @ W.double().t()).abs().max().item()
    assert (plain - A.double() @ W.double().t()).abs().max().item() < 4e-5 * scale      # fp32 MFMA: ~2e-6; plain bf16: ~1e-2
    assert (out - ref).abs().max().item() < 4e-5 * scale + 2e-6


def test_tile_choice_changes_no_bit(x3):
    g = torch.Generator().manual_seed(9)
    A, W = torch.randn(520, 1536, generator=g).cuda(), (torch.randn(512, 1536, generator=g) * 0.03).cuda()
    outs = [x3.diag_gemm(A, W, tile=t) for t in TILES]
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    x3.gemm_precision = "fp32"
    f32 = x3.diag_gemm(A, W, tile=37)
    assert not torch.equal(f32, outs[0]) and (f32 - outs[0]).abs().max().item() < 5e-5 * f32.abs().max().item()


def test_scoring_forward_small_batch_vs_oracle(x3, sd0):
    """embed() on fp32 buffers with three-product GEMMs: embeddings within 2e-5 of the CPU oracle, scores within 1e-4."""
    gen = torch.Generator().manual_seed(3)
    wav = (0.1 * torch.randn(5, 24000, generator=gen)).clamp(-1, 1)
    with torch.no_grad():
        ref = O.triplet_forward(sd0, wav)
    emb = x3.embed(wav.cuda())
    assert (emb.cpu() - ref).abs().max().item() < 2e-5
    d, m = x3.pairwise(emb[:2].contiguous(), emb[2:].contiguous())
    dref, mref = O.pairwise(ref[:2].numpy(), ref[2:].numpy())
    assert np.abs(d.cpu().numpy() - dref).max() < 1e-4 and np.abs(m.cpu().numpy() - mref).max() < 1e-4
    single = x3.embed(wav[3:4].contiguous().cuda())          # batch invariance holds in this mode too
    assert torch.equal(single[0], emb[3])


def _oracle_grad(sd, wav, head_w, head_b, G_layers, G_emb, mult):
    w = wav.clone().requires_grad_(True)
    outs = O.lossnet_forward(sd, w, head_w, head_b, feature_grad_mult=mult, required_seq_len_multiple=2)
    s = sum((outs[i] * G_layers[i]).sum() for i in range(12)) + (outs[12] * G_emb).sum()
    (grad,) = torch.autograd.grad(s, w)
    return grad


@pytest.mark.parametrize("mult", [1.0, 0.1])
def test_full_chain_backward_vs_oracle_autograd(x3, sd0, mult):
    """The gate of tests/test_gpu_backward.py::test_full_chain_backward_over_three_attention_tiles, in this mode: a smooth
    functional of the 13 outputs of a 42 000-sample clip (T = 131) - rel < 1e-3, cosine > 0.999999 vs the oracle's autograd,
    and vs the HF + GradMultiply golden."""
    default = x3.feature_grad_mult
    x3.feature_grad_mult = mult
    try:
        g = np.load(os.path.join(GOLD, "hf_grad_fgm.npz"))
        l = np.load(os.path.join(GOLD, "hf_loss.npz"))
        wav = torch.from_numpy(g["long_wav"])
        hw, hb = torch.from_numpy(l["emb_w"]), torch.from_numpy(l["emb_b"])
        gen = torch.Generator().manual_seed(int(g["g_seed"]))
        T = 131
        G_layers = torch.randn(12, 1, T, 768, generator=gen) / (T * 768)
        G_emb = torch.randn(1, 256, generator=gen) / 256
        ref = _oracle_grad(sd0, wav, hw, hb, G_layers, G_emb, mult)
        head = (hw.cuda(), hb.cuda())
        emb, layers, saved = x3.embed_train(wav.cuda(), head)
        dwav = x3.embed_backward(wav.cuda(), layers, saved, G_layers.cuda(), G_emb.cuda(), head).cpu()
        rel = (dwav - ref).abs().max().item() / ref.abs().max().item()
        assert rel < 1e-3, rel
        assert F.cosine_similarity(dwav.flatten(), ref.flatten(), dim=0).item() > 0.999999
        hf = torch.from_numpy(g["long_grad_fgm01" if mult == 0.1 else "long_grad_fgm1"])
        assert (dwav - hf).abs().max().item() / hf.abs().max().item() < 1e-3
    finally:
        x3.feature_grad_mult = default


def test_forward_loss_and_gradient_vs_fp32_path_and_hf_golden(built_lib, sd0):
    """Nomad(precision="bf16x3").forward() on the hf_loss.npz inputs: the loss to 1e-5 (relative) of the golden, the L1
    gradient (piecewise constant in the sign of est - clean) within the tolerance the fp32 path is held to."""
    from nomad_amd.nomad import Nomad
    g = np.load(os.path.join(GOLD, "hf_loss.npz"))
    gf = np.load(os.path.join(GOLD, "hf_grad_fgm.npz"))
    est0, clean = torch.from_numpy(g["estimate"]).cuda(), torch.from_numpy(g["clean"]).cuda()
    res = {}
    for prec in ("fp32", "bf16x3"):
        nmd = Nomad(weights=sd0, precision=prec)
        assert nmd.engine.gemm_precision == prec
        nmd.lossnet_layers.embedding_weight = torch.from_numpy(g["emb_w"]).cuda()
        nmd.lossnet_layers.embedding_bias = torch.from_numpy(g["emb_b"]).cuda()
        est = est0.clone().requires_grad_(True)
        loss = nmd.forward(est, clean)
        loss.backward()
        res[prec] = (float(loss), est.grad.cpu())
        nmd.engine.close()
    assert abs(res["bf16x3"][0] - float(g["loss"])) < 1e-4 * float(g["loss"])
    assert abs(res["bf16x3"][0] - res["fp32"][0]) < 2e-5 * res["fp32"][0]
    ref = torch.from_numpy(gf["grad_l1_fgm01"])
    scale = ref.abs().max().item()
    assert (res["bf16x3"][1] - ref).abs().max().item() < 3e-3 * scale
    assert (res["bf16x3"][1] - res["fp32"][1]).norm().item() < 2e-3 * res["fp32"][1].norm().item()


def test_training_step_parameter_gradients_vs_oracle_autograd(built_lib):
    """The triplet fine-tuning step (train_triplet.py:117-131) with gemm_precision = "bf16x3": forward, dX and the split-K dW
    GEMMs on three bf16 products, fp32 accumulation - every parameter gradient against torch.autograd on the CPU oracle
    (per tensor 1e-3 of its largest gradient + 1e-5 of the largest anywhere; the fp32 mode holds 2e-4 / 1e-6), cosine > 0.99999
    over the whole vector; loss to 1e-4."""
    from nomad_amd.engine import Engine
    from nomad_amd.weights import seeded_state_dict
    sd = seeded_state_dict(3, qk_gain=3.0)
    eng = Engine({k: v.clone() for k, v in sd.items()}, 0)
    eng.gemm_precision = "bf16x3"
    eng.train_enable()
    g = torch.Generator().manual_seed(2)
    A, P, N = [(0.1 * torch.randn(2, 48000, generator=g)).clamp(-1, 1) for _ in range(3)]     # T = 149: three attention tiles
    ref_loss, ref = O.triplet_step_grads(sd, A, P, N, 1.0)
    eng.train_zero_grad()
    outs = [eng.embed_train(w.cuda()) for w in (A, P, N)]
    loss, da, dp, dn = eng.triplet_loss(outs[0][0], outs[1][0], outs[2][0], 1.0)
    for w, (emb, layers, saved), d in zip((A, P, N), outs, (da, dp, dn)):
        eng.train_backward(w.cuda(), layers, saved, d)
    flat = eng.train_read(1)
    got = eng.train_unflatten(flat)
    assert abs(loss.item() - ref_loss.item()) < 1e-4
    top = max(v.abs().max().item() for v in ref.values())
    worst = ("", 0.0)
    for k, want in ref.items():
        err = (got[k] - want).abs().max().item() / (1e-3 * want.abs().max().item() + 1e-5 * top)
        if err > worst[1]:
            worst = (k, err)
    assert worst[1] < 1.0, worst
    w = torch.cat([(ref[k] if k in ref else torch.zeros(n)).reshape(-1) for k, _, n in eng.train_segments()]).double()
    gflat = flat.cpu().double()
    assert (w @ gflat / (w.norm() * gflat.norm())).item() > 0.99999
    eng.close()
