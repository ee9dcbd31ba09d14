"""GPU: the bf16x3 path - fp32-class results from the bf16 matrix cores.  GEMM operands are split into hi / lo bf16
planes and multiplied as hi*hi + hi*lo + lo*hi with fp32 accumulation (gemm_bf16_8phase.hip.h, X3).  Checked against
float64 references and this library's fp32 path; the bar is the north star's score tolerance (1e-4) with a wide margin."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def test_split_round_trip(engine):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1000, 768, generator=g) * torch.logspace(-6, 6, 768)
    xs = engine.diag_split_bf16(x.cuda())
    hi, lo = xs[0].float().cpu(), xs[1].float().cpu()
    assert torch.equal(hi, x.bfloat16().float())                       # hi = RNE bf16 of x
    assert torch.equal(lo, (x - hi).bfloat16().float())                # lo = RNE bf16 of the remainder
    back = engine.diag_unsplit_bf16(xs).cpu()
    assert ((back - x).abs() <= x.abs() * 2.0 ** -16).all()            # 16+ significant bits survive
    assert torch.equal(back, hi + lo)


# the two GEMM kernels of the path: K-concatenated operands in the 8-phase bf16 kernel (variants 0 split / 1 fp32 out) and
# the kernel that stages every plane once (gemm_bf16x3.hip.h, variants 7 / 8)
KERNELS = {"kcat": (0, 1), "staged": (7, 8), "staged3": (12, 13)}   # staged3: three A buffers, K % 192 == 0


@pytest.mark.parametrize("kernel", sorted(KERNELS))
@pytest.mark.parametrize("M", [1, 200, 257, 1000])
def test_gemm_bf16x3_exact_on_16_bit_integers(engine_for, M, kernel):
    """Operands that hi + lo represents exactly (|a| < 2^16) whose lo*lo terms vanish (one operand has lo = 0):
    the three products reproduce the exact integer result as long as it fits fp32."""
    N, K = 512, (384 if kernel == "staged3" else 256)
    var = KERNELS[kernel][1]
    engine = engine_for("bf16x3", var)
    g = torch.Generator().manual_seed(M)
    A = torch.randint(-20000, 20000, (M, K), generator=g).float()       # needs hi and lo
    W = torch.randint(-3, 4, (N, K), generator=g).float()               # lo plane is zero
    W[:, ::7] = 1.0
    ref = A.double() @ W.double().T
    assert ref.abs().max() < 2 ** 24
    out = engine.diag_gemm_bf16x3(engine.diag_split_bf16(A.cuda()), engine.diag_split_bf16(W.cuda()), variant=var).cpu()
    assert torch.equal(out.double(), ref)
    # and with the roles swapped: the lo plane on the W side
    A2 = torch.randint(-3, 4, (M, K), generator=g).float()
    W2 = torch.randint(-20000, 20000, (N, K), generator=g).float()
    out2 = engine.diag_gemm_bf16x3(engine.diag_split_bf16(A2.cuda()), engine.diag_split_bf16(W2.cuda()), variant=var).cpu()
    assert torch.equal(out2.double(), A2.double() @ W2.double().T)


@pytest.mark.parametrize("kernel", sorted(KERNELS))
@pytest.mark.parametrize("M,N,K", [(777, 768, 3072), (1500, 512, 1536), (300, 2304, 768), (4113, 256, 128), (600, 3072, 768),
                                   (513, 256, 64), (255, 512, 192 + 64)])
@pytest.mark.parametrize("epi", ["none", "bias_gelu", "bias_res"])
@pytest.mark.parametrize("out_f32", [True, False])
def test_gemm_bf16x3_vs_float64(engine_for, M, N, K, epi, out_f32, kernel):
    if kernel == "kcat" and K % 128:
        pytest.skip("the K-concatenated kernel walks K tiles of 64 in pairs")
    if kernel == "staged3" and K % 192:
        pytest.skip("three A buffers: stages in sixes")
    engine = engine_for("bf16x3", KERNELS[kernel][1])
    g = torch.Generator().manual_seed(5)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * K ** -0.5
    bias = torch.randn(N, generator=g) if "bias" in epi else None
    R = torch.randn(M, N, generator=g) if "res" in epi else None
    ref = A.double() @ W.double().T
    if bias is not None:
        ref = ref + bias.double()
    if "gelu" in epi:
        ref = F.gelu(ref)
    if R is not None:
        ref = ref + R.double()
    out = engine.diag_gemm_bf16x3(engine.diag_split_bf16(A.cuda()), engine.diag_split_bf16(W.cuda()),
                                  bias.cuda() if bias is not None else None,
                                  engine.diag_split_bf16(R.cuda()) if R is not None else None, gelu="gelu" in epi,
                                  variant=KERNELS[kernel][1 if out_f32 else 0])
    if not out_f32:
        out = engine.diag_unsplit_bf16(out)
    err = (out.cpu().double() - ref).abs().max().item()
    f32 = (A.cuda() @ W.cuda().T).cpu().double()
    if epi == "none":
        print(f"bf16x3 {kernel} {M}x{N}x{K}: max|err| {err:.2e}  (torch fp32 matmul: {(f32 - (A.double() @ W.double().T)).abs().max().item():.2e})")
    # per product 2^-16 relative (dropped lo*lo + two 2^-17 operand roundings), random signs over K, |a||w| ~ K^-1/2;
    # measured ~3e-6 at unit-scale outputs; a plain bf16 GEMM is ~1e-2 here
    assert err < 3e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("B,T", [(2, 50), (1, 64), (1, 65), (3, 199), (1, 330), (1, 1), (2, 256), (1, 225), (1, 17)])
@pytest.mark.parametrize("gain", [1.0, 6.0, 40.0])
def test_attention_bf16x3(engine, B, T, gain):
    """Split-operand attention against float64: logits up to a few hundred (gain 40) still come out to ~1e-5."""
    g = torch.Generator().manual_seed(T)
    qkv = torch.randn(B * T, 2304, generator=g)
    qkv[:, :1536] *= (gain ** 0.5) * 0.35
    q, k, v = (qkv[:, i * 768:(i + 1) * 768].double().view(B, T, 12, 64).transpose(1, 2) for i in range(3))
    ref = (torch.softmax(q @ k.transpose(-1, -2), -1) @ v).transpose(1, 2).reshape(B * T, 768)
    qs = engine.diag_split_bf16(qkv.cuda())
    tiled = engine.diag_attention_bf16x3(qs, B, T, waves=0)
    out = engine.diag_unsplit_bf16(tiled).cpu()
    assert torch.isfinite(out).all()
    if T <= 256:   # the K/V-resident kernel walks the same tiles in the same order: same bits
        for waves in (4, 8):
            assert torch.equal(engine.diag_attention_bf16x3(qs, B, T, waves=waves), tiled), waves
    else:
        with pytest.raises(Exception):
            engine.diag_attention_bf16x3(qs, B, T, waves=8)
    err = (out.double() - ref).abs().max().item()
    f32 = engine.diag_attention(qkv.cuda(), B, T).cpu() if hasattr(engine, "diag_attention") else None
    e32 = (f32.double() - ref).abs().max().item() if f32 is not None else float("nan")
    print(f"attention bf16x3 B={B} T={T} gain={gain}: max|err| {err:.2e} (fp32 kernel {e32:.2e})")
    # the split products put ~2^-16 relative noise on the logits: 15-20x the fp32 kernel's error, which grows with the logits too
    assert err < max(2e-5, 30 * e32) * max(1.0, ref.abs().max().item()), err


def test_embed_bf16x3_vs_fp32_path(engine):
    gen = torch.Generator().manual_seed(0)
    wav = (0.1 * torch.randn(8, 64000, generator=gen)).clamp(-1, 1).cuda()
    e32 = engine.embed(wav)
    ex3 = engine.embed_bf16x3(wav)
    torch.cuda.synchronize()
    assert torch.isfinite(ex3).all()
    assert (ex3.norm(dim=1) - 1).abs().max().item() < 1e-5
    err = (ex3 - e32).abs().max().item()
    d32, m32 = engine.pairwise(e32[:6].contiguous(), e32[6:].contiguous())
    dx3, mx3 = engine.pairwise(ex3[:6].contiguous(), ex3[6:].contiguous())
    serr = (dx3 - d32).abs().max().item()
    print(f"bf16x3 vs fp32: embedding max|err| {err:.3e}, score max|err| {serr:.3e}")
    assert err < 1e-5 and serr < 2e-5          # the north star's score tolerance is 1e-4


def test_embed_bf16x3_vs_oracle_scores(engine, sd0):
    """Against the CPU oracle (fp32 torch + float64 distances): inside the 1e-4 score tolerance like the fp32 path."""
    from oracle import nomad_oracle as O
    gen = torch.Generator().manual_seed(3)
    wav = (0.1 * torch.randn(5, 16000, generator=gen)).clamp(-1, 1)
    ref = O.triplet_forward(sd0, wav)
    ex3 = engine.embed_bf16x3(wav.cuda()).cpu()
    assert (ex3 - ref).abs().max().item() < 2e-5
    d_ref, m_ref = O.pairwise(ref[:3].numpy(), ref[3:].numpy())
    d, m = O.pairwise(ex3[:3].numpy(), ex3[3:].numpy())
    assert abs(d - d_ref).max() < 1e-4 / 4


@pytest.mark.parametrize("n_samples", [400, 16000, 16080, 16400, 16720, 17040, 17360, 40000])
def test_embed_bf16x3_frame_counts(engine, n_samples):
    """Frame counts in every residue class modulo the pos-conv's 5-frame blocks (T = 1, 49, 50, 51, 52, 53, 54, 124):
    the last block's surplus frames must not be stored, its reads past the clip's padding must be harmless."""
    gen = torch.Generator().manual_seed(n_samples)
    wav = (0.1 * torch.randn(3, n_samples, generator=gen)).clamp(-1, 1).cuda()
    e32 = engine.embed(wav)
    ex3 = engine.embed_bf16x3(wav)
    assert torch.isfinite(ex3).all()
    assert (ex3 - e32).abs().max().item() < (1e-4 if n_samples == 400 else 1e-5)   # one frame: no time averaging


def test_embed_bf16x3_batch_invariance_and_repeat(engine):
    gen = torch.Generator().manual_seed(4)
    wav = (0.1 * torch.randn(6, 40000, generator=gen)).clamp(-1, 1).cuda()
    a = engine.embed_bf16x3(wav)
    b = engine.embed_bf16x3(wav)
    assert torch.equal(a, b)
    one = engine.embed_bf16x3(wav[2:3].contiguous())
    assert torch.equal(one[0], a[2])


def test_embed_bf16x3_peaky_weights(engine_peaky):
    """Weights with large attention logits and activation outliers (the conftest 'peaky' model)."""
    gen = torch.Generator().manual_seed(6)
    wav = (0.3 * torch.randn(4, 32000, generator=gen)).clamp(-1, 1).cuda()
    e32 = engine_peaky.embed(wav)
    ex3 = engine_peaky.embed_bf16x3(wav)
    err = (ex3 - e32).abs().max().item()
    print(f"bf16x3 vs fp32 (peaky): embedding max|err| {err:.3e}")
    assert err < 5e-5


def test_ragged_bf16x3_bit_identical_to_single_clips(engine):
    """nomad_embed_ragged_bf16x3: clips of different lengths in one launch sequence, each bit-equal to its own
    nomad_embed_bf16x3 call and within 1e-5 of the fp32 ragged path."""
    g = torch.Generator().manual_seed(21)
    lens = [16384, 400, 27225, 9001, 64000, 30267, 5000, 12345, 48000]
    waves = [(0.1 * torch.randn(n, generator=g)).clamp(-1, 1) for n in lens]
    rag = engine.embed_ragged(waves, precision="bf16x3")
    for i, w in enumerate(waves):
        single = engine.embed_bf16x3(w[None, :].cuda())
        assert torch.equal(rag[i], single[0]), (i, lens[i])
    f32 = engine.embed_ragged(waves)
    assert (rag - f32).abs().max().item() < 1e-5
    with pytest.raises(ValueError):
        engine.embed_ragged(waves, head=(torch.zeros(256, 768).cuda(), torch.zeros(256).cuda()), precision="bf16x3")
    with pytest.raises(ValueError):
        engine.embed_ragged(waves, precision="fp16")


def test_predict_in_bf16x3_precision(built_lib):
    """Nomad(precision='bf16x3').predict on the reference's example files: the 3-decimal tables the reference prints and
    saves (nomad.py:115-117) come out identical to the fp32 run's, the raw scores agree to 1e-5."""
    import os
    import tempfile
    from conftest import GOLD
    from nomad_amd.nomad import Nomad
    nmr, deg = os.path.join(GOLD, "wavs", "nmr-data"), os.path.join(GOLD, "wavs", "test-data")
    with tempfile.TemporaryDirectory() as d:
        a32, m32 = Nomad(weights="seeded").predict("dir", nmr, deg, results_path=d)
        ax3, mx3 = Nomad(weights="seeded", precision="bf16x3").predict("dir", nmr, deg, results_path=d)
    assert list(mx3.columns) == list(m32.columns) and list(mx3.index) == list(m32.index)
    assert abs(mx3.values - m32.values).max() <= 1e-3 + 1e-9          # at most one unit of the 3rd decimal (rounding edge)
    assert abs(ax3.values[:, -1].astype(float) - a32.values[:, -1].astype(float)).max() <= 1e-3 + 1e-9


def test_predict_bf16x3_small_batches_stay_on_fp32_buffers(built_lib, monkeypatch):
    """Six example files are far below BF16X3_MIN_SAMPLES: Nomad(precision='bf16x3') embeds them on fp32 buffers with
    three-product GEMMs (Engine.gemm_precision = "bf16x3": close to the fp32 embeddings, bit-equal to that engine mode); with
    the threshold lowered the same call runs the split-storage kernels (close to both, equal to neither)."""
    import os
    import numpy as np
    from conftest import GOLD
    import importlib
    NM = importlib.import_module("nomad_amd.nomad")
    nmr = os.path.join(GOLD, "wavs", "nmr-data")
    n32 = NM.Nomad(weights="seeded")
    e32 = n32.get_embeddings(nmr).iloc[:, 1:].values.astype(np.float32)
    nx3 = NM.Nomad(weights="seeded", precision="bf16x3")
    assert nx3.engine.gemm_precision == "bf16x3" and n32.engine.gemm_precision == "fp32"
    ex3 = nx3.get_embeddings(nmr).iloc[:, 1:].values.astype(np.float32)
    assert not np.array_equal(e32, ex3) and np.abs(e32 - ex3).max() < 1e-5
    n32.engine.gemm_precision = "bf16x3"                      # the same engine mode by hand: the same bits
    assert np.array_equal(n32.get_embeddings(nmr).iloc[:, 1:].values.astype(np.float32), ex3)
    monkeypatch.setattr(NM, "BF16X3_MIN_SAMPLES", 0)
    ex3b = nx3.get_embeddings(nmr).iloc[:, 1:].values.astype(np.float32)
    assert not np.array_equal(ex3, ex3b) and np.abs(e32 - ex3b).max() < 1e-5


def test_bf16x3_follows_weight_updates(built_lib, sd0):
    """The split weight copies are rebuilt after the master weights change (fine-tuning, nomad_train_write)."""
    from nomad_amd.engine import Engine
    eng = Engine(sd0, 0)
    gen = torch.Generator().manual_seed(8)
    wav = (0.1 * torch.randn(2, 20000, generator=gen)).cuda()
    before = eng.embed_bf16x3(wav).clone()
    eng.train_enable()
    theta = eng.train_read()
    eng.train_write(0, theta * 1.01)
    after = eng.embed_bf16x3(wav)
    ref = eng.embed(wav)
    assert not torch.equal(before, after)
    assert (after - ref).abs().max().item() < 1e-5


# ---- LossNetLayers outputs on the bf16x3 path (nomad_embed_layers_bf16x3): the no-gradient branch of nomad.forward() ----
@pytest.mark.parametrize("B,N", [(2, 6000), (3, 16400), (8, 64000)])
def test_layer_outputs_bf16x3_vs_fp32_path(engine, B, N):
    gen = torch.Generator().manual_seed(B * 1000 + N)
    wav = (0.1 * torch.randn(B, N, generator=gen)).clamp(-1, 1).cuda()
    hw = (torch.randn(256, 768, generator=gen) / 768 ** 0.5).cuda()
    hb = (0.01 * torch.randn(256, generator=gen)).cuda()
    e32, l32 = engine.embed(wav, head=(hw, hb), want_layers=True)
    ex3, lx3 = engine.embed_bf16x3(wav, head=(hw, hb), want_layers=True)
    torch.cuda.synchronize()
    assert lx3.shape == l32.shape and ex3.shape == e32.shape
    rel = [((lx3[l] - l32[l]).abs().max() / l32[l].abs().max()).item() for l in range(12)]
    print(f"bf16x3 layer outputs B={B} N={N}: max|err|/max|x| per layer {max(rel):.2e} (first {rel[0]:.2e}, last {rel[-1]:.2e}), "
          f"emb {(ex3 - e32).abs().max().item():.2e}")
    assert max(rel) < 5e-5 and (ex3 - e32).abs().max().item() < 1e-5     # measured 2.3e-5 / 1.9e-6
    # the layer-output variant is the scoring forward plus extra stores: same embedding bits with the checkpoint's head
    assert torch.equal(engine.embed_bf16x3(wav, want_layers=True)[0], engine.embed_bf16x3(wav))
    # the last layer output is what the head pools
    pooled = torch.relu(lx3[11].mean(dim=1)) @ hw.T + hb
    assert (torch.nn.functional.normalize(pooled, dim=1) - ex3).abs().max().item() < 1e-5


def test_layer_outputs_bf16x3_vs_oracle(engine, sd0):
    from oracle import nomad_oracle as O
    gen = torch.Generator().manual_seed(11)
    wav = (0.1 * torch.randn(2, 6000, generator=gen)).clamp(-1, 1)
    hw = torch.randn(256, 768, generator=gen) / 768 ** 0.5
    hb = 0.01 * torch.randn(256, generator=gen)
    with torch.no_grad():
        ref = O.lossnet_forward(sd0, wav, hw, hb)
    emb, layers = engine.embed_bf16x3(wav.cuda(), head=(hw.cuda(), hb.cuda()), want_layers=True)
    rel = max(((layers[l].cpu() - ref[l]).abs().max() / ref[l].abs().max()).item() for l in range(12))
    print(f"bf16x3 layer outputs vs oracle: {rel:.2e}, emb {(emb.cpu() - ref[12]).abs().max().item():.2e}")
    assert rel < 5e-5 and (emb.cpu() - ref[12]).abs().max().item() < 2e-5


@pytest.mark.parametrize("reps", [1, 2], ids=["c4_shape_fp32_buffers", "64_clips_split_storage_clean_branch"])
def test_forward_loss_bf16x3_clean_branch(built_lib, sd0, reps):
    """nomad.forward() with precision="bf16x3": every GEMM of both branches and of the backward on three bf16 products; the
    clean branch (no gradient) on the split-storage forward once the batch is large enough; loss within 1e-5 (relative) of
    the fp32 engine's, gradient w.r.t. estimate too."""
    from nomad_amd import nomad as NM
    import importlib
    NM = importlib.import_module("nomad_amd.nomad")
    gen = torch.Generator().manual_seed(5)
    clean = (0.1 * torch.randn(32 * reps, 1, 16384, generator=gen)).clamp(-1, 1).cuda()
    est = (clean + 0.02 * torch.randn(32 * reps, 1, 16384, generator=gen).cuda()).clamp(-1, 1)
    n32, nx3 = NM.Nomad(weights=sd0), NM.Nomad(weights=sd0, precision="bf16x3")
    nx3.lossnet_layers.embedding_weight = n32.lossnet_layers.embedding_weight
    nx3.lossnet_layers.embedding_bias = n32.lossnet_layers.embedding_bias
    # 32 x 16384 samples (1 600 frames) per branch stay on fp32 buffers with three-product GEMMs; from ~2 500 frames on the
    # no-gradient branch takes the split-storage forward
    assert NM._takes_bf16x3("bf16x3", clean) == (reps == 2) and not NM._takes_bf16x3("fp32", clean)
    out = {}
    for name, n in (("fp32", n32), ("bf16x3", nx3)):
        e = est.clone().requires_grad_(True)
        loss = n.forward(e, clean)
        loss.backward()
        out[name] = (loss.item(), e.grad.clone())
        with torch.no_grad():
            out[name + "_nograd"] = n.forward(est, clean).item()
    l32, g32 = out["fp32"]
    lx3, gx3 = out["bf16x3"]
    print(f"forward() loss fp32 {l32:.7f} bf16x3-clean {lx3:.7f} (rel {abs(lx3 - l32) / l32:.2e}); no-grad both branches bf16x3 "
          f"{out['bf16x3_nograd']:.7f}; grad rel {(gx3 - g32).abs().max().item() / g32.abs().max().item():.2e}")
    assert abs(lx3 - l32) / l32 < 1e-5
    assert abs(out["bf16x3_nograd"] - out["fp32_nograd"]) / l32 < 1e-5
    # d|e - c|/de = sign(e - c) flips wherever the clean branch moved by more than |e - c|: a few elements per layer
    l2 = ((gx3 - g32).norm() / g32.norm()).item()
    print(f"grad: relative L2 difference {l2:.2e}")
    assert (gx3 - g32).abs().max().item() / g32.abs().max().item() < 1e-2 and l2 < 1e-2
    # small batches stay on fp32 buffers, with three-product GEMMs in this mode: close, not bit-equal
    a = n32.forward(est[:4], clean[:4]).item()
    b = nx3.forward(est[:4], clean[:4]).item()
    assert abs(a - b) / a < 1e-5
    # "not bit-equal" is checked on the embeddings, not on the loss: the two losses are about one fp32 ulp apart and the head is
    # initialised at random per process, so they round to the same float in a good share of the runs (seen on the GPU box)
    assert n32.engine.gemm_precision == "fp32" and nx3.engine.gemm_precision == "bf16x3"
    ea, eb = n32.engine.embed(est[:4].squeeze(1)), nx3.engine.embed(est[:4].squeeze(1))
    assert not torch.equal(ea, eb) and (ea - eb).abs().max().item() < 1e-4
