"""GPU: each HIP kernel, called through the C ABI (libnomad_hip.so), against a CPU reference."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# nomad_diag_gemm tile ids -> (BM, BN, BK); >= 20 are the LDS-DMA (global_load_lds) instantiations
# 20 / 31 / 33 / 34 / 37 are what the product selects (libnomad_hip.so); the others are experiments (libnomad_diag.so)
TILES = {0: (128, 128, 32), 1: (128, 64, 16), 2: (64, 64, 32), 6: (256, 128, 16),
         20: (128, 128, 32), 21: (256, 128, 16), 22: (128, 128, 16), 28: (128, 64, 16), 29: (128, 64, 32), 31: (128, 128, 32),
         33: (256, 128, 16), 34: (128, 64, 32), 35: (256, 128, 32), 37: (64, 64, 32)}


def _dev(x):
    return x.to("cuda").contiguous()


@pytest.mark.parametrize("tile", sorted(TILES))
@pytest.mark.parametrize("M", [1, 63, 200, 257, 1000])
def test_gemm_exact_integer_asymmetric(engine_for, tile, M):
    """Exact small-integer operands with an asymmetric W: any MFMA operand/output layout slip
    (row<->col swap, wrong k pairing) shows up as a hard mismatch."""
    bm, bn, bk = TILES[tile]
    N, K = 2 * bn, 3 * bk * 2
    g = torch.Generator().manual_seed(M * 7 + tile)
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    W = torch.randint(-3, 4, (N, K), generator=g).float()
    W += (torch.arange(N)[:, None] % 5).float() - (torch.arange(K)[None, :] % 3).float()  # break symmetry
    ref = (A.double() @ W.double().T).float()
    out = engine_for("f32", tile).diag_gemm(_dev(A), _dev(W), tile=tile).cpu()
    assert torch.equal(out, ref)


@pytest.mark.parametrize("tile,M,N,K", [(0, 500, 256, 768), (0, 130, 768, 3072), (2, 84, 768, 512),
                                        (2, 300, 512, 1536), (1, 260, 64, 96), (21, 1500, 256, 768),
                                        (29, 1100, 768, 3072), (28, 260, 64, 96), (21, 700, 384, 1536),
                                        (33, 1500, 256, 768), (34, 1100, 768, 3072), (33, 300, 128, 16),
                                        (34, 260, 64, 32), (33, 700, 384, 48), (37, 84, 768, 512), (37, 300, 512, 1536),
                                        (31, 500, 256, 768), (31, 130, 768, 3072), (20, 700, 2304, 768)])
@pytest.mark.parametrize("epi", ["none", "bias", "bias_gelu", "bias_res", "bias_gelu_res"])
def test_gemm_epilogues(engine_for, tile, M, N, K, epi):
    g = torch.Generator().manual_seed(11)
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
    out = engine_for("f32", tile).diag_gemm(_dev(A), _dev(W), _dev(bias) if bias is not None else None,
                                           _dev(R) if R is not None else None, gelu="gelu" in epi, tile=tile).cpu()
    err = (out.double() - ref).abs().max().item()
    assert err < 1e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("M,N,K,epi", [(25400, 768, 768, "bias_res"), (25472, 768, 3072, "bias_res"), (50900, 3072, 768, "bias_gelu"),
                                       (25472, 2304, 768, "bias"), (51000, 768, 768, "none")])
def test_gemm_two_tile_shapes_in_one_launch(engine, M, N, K, epi):
    """gemm_f32_mixed_kernel (what the product dispatch makes of tile 33 when the last round of 256 x 128 tiles would be sparsely
    filled: 256 x 128 tiles over the rows of the whole rounds, 128 x 128 tiles over the rest; M deliberately not always a multiple
    of 128) must equal the single-shape 128 x 128 x 32 kernel (tile 31, never split) bit for bit - every instantiation contracts
    k in the same order - and, on small integers, the exact product."""
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    W = torch.randint(-2, 3, (N, K), generator=g).float() + (torch.arange(N)[:, None] % 3).float()
    bias = torch.randint(-4, 5, (N,), generator=g).float() if "bias" in epi else None
    R = torch.randint(-8, 9, (M, N), generator=g).float() if "res" in epi else None
    args = (_dev(A), _dev(W), _dev(bias) if bias is not None else None, _dev(R) if R is not None else None)
    mixed = engine.diag_gemm(*args, gelu="gelu" in epi, tile=33)
    single = engine.diag_gemm(*args, gelu="gelu" in epi, tile=31)
    assert torch.equal(mixed, single)
    if "gelu" not in epi:   # |values| < 2^24: the fp32 result is the exact integer
        rows = torch.cat([torch.arange(0, 300), torch.arange(M - 300, M)])   # both parts of the launch
        ref = A[rows].double() @ W.double().T
        if bias is not None:
            ref = ref + bias.double()
        if R is not None:
            ref = ref + R[rows].double()
        assert torch.equal(mixed[rows.cuda()].cpu().double(), ref)
    # random data as well (rounding in every product)
    A2 = torch.randn(M, K, generator=g)
    W2 = torch.randn(N, K, generator=g) * K ** -0.5
    a2 = (_dev(A2), _dev(W2), args[2], args[3])
    assert torch.equal(engine.diag_gemm(*a2, gelu="gelu" in epi, tile=33), engine.diag_gemm(*a2, gelu="gelu" in epi, tile=31))


@pytest.mark.parametrize("M", [1, 255, 256, 1000, 1500])
@pytest.mark.parametrize("epi", ["exact", "bias_gelu_res"])
def test_gemm_n48_kernel(engine, M, epi):
    """The N = 48 instantiation (16x16x4 MFMA, used by the grouped pos-conv): exact integers + fused epilogue."""
    g = torch.Generator().manual_seed(M)
    K = 6144 if M == 1500 else 160
    if epi == "exact":
        A = torch.randint(-3, 4, (M, K), generator=g).float()
        W = torch.randint(-3, 4, (48, K), generator=g).float()
        W += (torch.arange(48)[:, None] % 5).float() - (torch.arange(K)[None, :] % 3).float()
        out = engine.diag_gemm(_dev(A), _dev(W), tile=48).cpu()
        assert torch.equal(out, (A.double() @ W.double().T).float())
    else:
        A = torch.randn(M, K, generator=g)
        W = torch.randn(48, K, generator=g) * K ** -0.5
        bias, R = torch.randn(48, generator=g), torch.randn(M, 48, generator=g)
        ref = F.gelu(A.double() @ W.double().T + bias.double()) + R.double()
        out = engine.diag_gemm(_dev(A), _dev(W), _dev(bias), _dev(R), gelu=True, tile=48).cpu()
        assert (out.double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("N", [512, 768])
@pytest.mark.parametrize("M", [1, 5, 199, 1030])
def test_layernorm(engine, M, N):
    g = torch.Generator().manual_seed(N + M)
    x = torch.randn(M, N, generator=g) * 3 + 0.7
    gamma, beta = 1 + 0.1 * torch.randn(N, generator=g), 0.1 * torch.randn(N, generator=g)
    ref = F.layer_norm(x.double(), (N,), gamma.double(), beta.double(), 1e-5)
    out = engine.diag_layernorm(_dev(x), _dev(gamma), _dev(beta)).cpu()
    assert (out.double() - ref).abs().max().item() < 1e-5


@pytest.mark.parametrize("B,T", [(2, 50), (1, 64), (2, 65), (1, 199), (1, 330)])
@pytest.mark.parametrize("gain", [1.0, 8.0])
def test_attention(engine, B, T, gain):
    """softmax(q k^T) v per head vs float64; gain 8 makes the softmax peaky (online-softmax rescale path)."""
    g = torch.Generator().manual_seed(T)
    qkv = torch.randn(B * T, 2304, generator=g)
    qkv[:, :1536] *= gain ** 0.5
    q, k, v = (qkv[:, i * 768:(i + 1) * 768].double().view(B, T, 12, 64).transpose(1, 2) for i in range(3))
    ref = (torch.softmax(q @ k.transpose(-1, -2), -1) @ v).transpose(1, 2).reshape(B * T, 768)
    out = engine.diag_attention(_dev(qkv), B, T).cpu()
    # tolerance = what plain fp32 arithmetic (torch CPU) achieves on the same data, x4, + 2e-6
    q32, k32, v32 = (qkv[:, i * 768:(i + 1) * 768].view(B, T, 12, 64).transpose(1, 2) for i in range(3))
    f32 = (torch.softmax(q32 @ k32.transpose(-1, -2), -1) @ v32).transpose(1, 2).reshape(B * T, 768)
    tol = 4 * (f32.double() - ref).abs().max().item() + 2e-6
    assert (out.double() - ref).abs().max().item() < tol


def test_attention_forced_rescale(engine):
    """A key in a LATER tile dominates one query row: the running max must jump and rescale O, l."""
    B, T = 1, 199
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(T, 2304, generator=g) * 0.3
    qkv[7, 0:64] = 2.0           # query row 7, head 0
    qkv[150, 768:832] = 2.0      # key 150 (third tile) aligned with it -> score 256
    q, k, v = (qkv[:, i * 768:(i + 1) * 768].double().view(B, T, 12, 64).transpose(1, 2) for i in range(3))
    ref = (torch.softmax(q @ k.transpose(-1, -2), -1) @ v).transpose(1, 2).reshape(T, 768)
    out = engine.diag_attention(_dev(qkv), B, T).cpu()
    assert torch.isfinite(out).all()
    assert (out.double() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("Nd,Nr", [(2, 4), (70, 130), (33, 64), (300, 1000)])
def test_pairwise_matches_scipy(engine, Nd, Nr):
    from scipy.spatial.distance import cdist
    rng = np.random.default_rng(Nd)
    a = rng.standard_normal((Nd, 256)).astype(np.float32)
    b = rng.standard_normal((Nr, 256)).astype(np.float32)
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    b[1] = a[0]                                                            # exact duplicate -> 0
    b[2] = a[1] + 1e-4 * rng.standard_normal(256).astype(np.float32)       # small-distance regime
    d, m = engine.pairwise(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
    ref = cdist(a, b)
    assert d.dtype == torch.float64
    assert np.abs(d.cpu().numpy() - ref).max() < 1e-13
    assert d[0, 1].item() == 0.0
    assert np.abs(m.cpu().numpy() - ref.mean(axis=1)).max() < 1e-13
    _, m2 = engine.pairwise(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), want_matrix=False)
    assert torch.equal(m, m2)  # deterministic, matrix optional


def test_l1_loss(engine):
    g = torch.Generator().manual_seed(0)
    B, T = 3, 50
    a, b = torch.randn(12, B, T, 768, generator=g), torch.randn(12, B, T, 768, generator=g)
    ea, eb = torch.randn(B, 256, generator=g), torch.randn(B, 256, generator=g)
    ref = sum(F.l1_loss(a[i].double(), b[i].double()) for i in range(12)) + F.l1_loss(ea.double(), eb.double())
    out = engine.l1_loss(_dev(a), _dev(b), _dev(ea), _dev(eb))
    assert abs(out.item() - ref.item()) < 1e-5
