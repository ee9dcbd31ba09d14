"""GPU: the whole hot path through the C ABI vs the oracle and the committed golden vectors.

Tolerances (fp32; BASELINE.json north star: per-pair NOMAD scores within 1e-4 of the reference):
  intermediate activations / layer outputs  <= 1e-4 abs (values reach ~5)
  embeddings (unit norm)                    <= 1e-5 abs
  distances / NOMAD scores                  <= 1e-4 abs
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLD
from oracle import nomad_oracle as O

pytestmark = pytest.mark.gpu
LAYER_TOL, EMB_TOL, SCORE_TOL = 1e-4, 1e-5, 1e-4


def test_tiny_batch_vs_hf_golden(engine):
    g = np.load(os.path.join(GOLD, "hf_tiny.npz"))
    wav = torch.from_numpy(g["wav"]).cuda()
    emb, layers = engine.embed(wav, want_layers=True)
    torch.cuda.synchronize()
    assert layers.shape == (12, 3, 18, 768)
    for l in range(12):
        err = (layers[l].cpu() - torch.from_numpy(g["layers"][l])).abs().max().item()
        assert err < LAYER_TOL, (l, err)
    assert (emb.cpu() - torch.from_numpy(g["emb"])).abs().max().item() < EMB_TOL


def test_peaky_attention_vs_hf_golden(engine_peaky):
    g = np.load(os.path.join(GOLD, "hf_tiny_peaky.npz"))
    emb, layers = engine_peaky.embed(torch.from_numpy(g["wav"]).cuda(), want_layers=True)
    assert (layers[0].cpu() - torch.from_numpy(g["layer0"])).abs().max().item() < LAYER_TOL
    assert (layers[11].cpu() - torch.from_numpy(g["last"])).abs().max().item() < 5e-4
    assert (emb.cpu() - torch.from_numpy(g["emb"])).abs().max().item() < 5e-5


def test_every_stage_vs_oracle(engine, sd0):
    """Intermediates of one small ragged batch: localises a failure to a kernel."""
    gen = torch.Generator().manual_seed(42)
    B, N = 2, 9000
    wav = (0.1 * torch.randn(B, N, generator=gen)).clamp(-1, 1)
    taps = {}
    with torch.no_grad():
        O.backbone(sd0, wav, taps)
    engine.diag_keep_intermediates(True)
    try:
        engine.embed(wav.cuda())
        torch.cuda.synchronize()
        T = taps["conv6"].shape[1]
        for i in range(7):
            got = engine.diag_region(B, N, f"conv{i}").cpu().view(B, -1, 512)
            err = (got - taps[f"conv{i}"]).abs().max().item()
            assert err < 2e-5, (f"conv{i}", err)
        xg = engine.diag_region(B, N, "xpad").cpu().view(16, B, T + 128, 48)   # group-major pos-conv input
        assert xg[:, :, :64].abs().max().item() == 0.0 and xg[:, :, 64 + T:].abs().max().item() == 0.0
        proj = xg[:, :, 64:64 + T].permute(1, 2, 0, 3).reshape(B, T, 768)
        assert (proj - taps["proj"]).abs().max().item() < 2e-5
    finally:
        engine.diag_keep_intermediates(False)


def test_batch_invariance_is_bit_exact(engine):
    """A clip's embedding must not depend on what else is in the batch (no padding, no cross-clip math)."""
    gen = torch.Generator().manual_seed(9)
    wav = (0.1 * torch.randn(5, 16384, generator=gen)).clamp(-1, 1).cuda()
    full = engine.embed(wav).clone()
    for i in (0, 3):
        one = engine.embed(wav[i:i + 1].contiguous())
        assert torch.equal(one[0], full[i])


def test_c2_shape_small_batch_vs_oracle(engine, sd0):
    """Config C2's clip shape (16 kHz x 4 s -> T=199) at B=6 against the CPU oracle."""
    gen = torch.Generator().manual_seed(0)
    wav = (0.1 * torch.randn(6, 64000, generator=gen)).clamp(-1, 1)
    with torch.no_grad():
        ref = O.triplet_forward(sd0, wav)
    emb = engine.embed(wav.cuda()).cpu()
    assert (emb - ref).abs().max().item() < EMB_TOL
    d, m = engine.pairwise(emb[:4].cuda().contiguous(), emb[4:].cuda().contiguous())
    dref, mref = O.pairwise(ref[:4].numpy(), ref[4:].numpy())
    assert np.abs(d.cpu().numpy() - dref).max() < SCORE_TOL
    assert np.abs(m.cpu().numpy() - mref).max() < SCORE_TOL


def test_predict_example_2x4(built_lib, sd0, tmp_path):
    """The reference's own example (config C1) through the drop-in Nomad.predict surface."""
    from nomad_amd.nomad import Nomad
    g = np.load(os.path.join(GOLD, "hf_example_wavs.npz"))
    nmd = Nomad(weights=sd0)
    df_avg, df_dm = nmd.predict("dir", os.path.join(GOLD, "wavs", "nmr-data"), os.path.join(GOLD, "wavs", "test-data"),
                                results_path=str(tmp_path))
    assert df_avg.index.name == "Test File" and list(df_avg.columns) == ["NOMAD"]
    assert sorted(df_dm.columns) == sorted(str(n) for n in g["nmr_names"])
    for i, dn in enumerate(g["deg_names"]):
        assert abs(df_avg.loc[str(dn), "NOMAD"] - round(float(g["mean"][i]), 3)) <= 1.01e-3
        for j, rn in enumerate(g["nmr_names"]):
            assert abs(df_dm.loc[str(dn), str(rn)] - round(float(g["dist"][i, j]), 3)) <= 1.01e-3
    assert os.path.isfile(tmp_path / "nomad_avg.csv") and os.path.isfile(tmp_path / "nomad_scores.csv")
    # unrounded parity: embeddings + distances at the north-star tolerance
    emb = nmd.get_embeddings(os.path.join(GOLD, "wavs", "test-data")).set_index("filename")
    for i, dn in enumerate(g["deg_names"]):
        row = emb.loc[os.path.join(GOLD, "wavs", "test-data", f"{dn}.wav")].to_numpy(dtype=np.float32)
        assert np.abs(row - g["deg_emb"][i]).max() < EMB_TOL
    with pytest.raises(Exception, match="is not valid. Valid values are dir and csv"):
        nmd.predict("zip", "a", "b")
    with pytest.raises(Exception, match="does not exist"):
        nmd.predict("dir", "/nonexistent/nmr", "/nonexistent/deg")


def test_forward_loss_vs_hf_golden(built_lib, sd0):
    from nomad_amd.nomad import Nomad
    g = np.load(os.path.join(GOLD, "hf_loss.npz"))
    nmd = Nomad(weights=sd0)
    nmd.lossnet_layers.embedding_weight = torch.from_numpy(g["emb_w"]).cuda()
    nmd.lossnet_layers.embedding_bias = torch.from_numpy(g["emb_b"]).cuda()
    outs = nmd.lossnet_layers(torch.from_numpy(g["estimate"]).cuda())
    assert len(outs) == 13 and outs[0].shape == (2, 50, 768) and outs[12].shape == (2, 256)
    loss = nmd.forward(torch.from_numpy(g["estimate"]).cuda(), torch.from_numpy(g["clean"]).cuda())
    assert loss.dim() == 0
    assert abs(loss.item() - float(g["loss"])) < 1e-4


def test_full_c2_batch_properties(engine):
    """BASELINE config C2 at full size (256 x 64000): size-independent properties."""
    gen = torch.Generator().manual_seed(0)
    wav = (0.1 * torch.randn(256, 64000, generator=gen)).clamp(-1, 1).cuda()
    emb = engine.embed(wav)
    torch.cuda.synchronize()
    assert torch.isfinite(emb).all()
    assert (emb.norm(dim=1) - 1).abs().max().item() < 1e-5
    # batch-invariant, bit-exact - also for the clips at the end of each half of the batch (Engine.embed runs it as two halves), whose
    # rows the two-shape GEMM launches (gemm_f32_mixed_kernel) give to 128 x 128 tiles instead of 256 x 128 ones
    for k in (200, 110, 127, 255):
        one = engine.embed(wav[k:k + 1].contiguous())
        assert torch.equal(one[0], emb[k]), k
    d, m = engine.pairwise(emb, emb)
    assert d.diagonal().abs().max().item() == 0.0   # d(a,a) = 0 exactly (difference form)
    assert (d - d.T).abs().max().item() == 0.0
    assert d.max().item() <= 2.0 + 1e-6
    assert (m - d.mean(dim=1)).abs().max().item() < 1e-12


def test_long_form_30s_fp32_vs_oracle(engine, sd0):
    """Config C5's shape (30 s at 16 kHz -> T = 1499, 24 key tiles per attention row) in fp32 vs the CPU oracle."""
    gen = torch.Generator().manual_seed(5)
    wav = (0.1 * torch.randn(1, 480000, generator=gen)).clamp(-1, 1)
    with torch.no_grad():
        x, layers = O.backbone(sd0, wav)
        ref = O.triplet_forward(sd0, wav)
    emb, lay = engine.embed(wav.cuda(), want_layers=True)
    assert lay.shape == (12, 1, 1499, 768)
    assert (lay[0].cpu() - layers[0]).abs().max().item() < LAYER_TOL
    assert (lay[11].cpu() - layers[11]).abs().max().item() < LAYER_TOL
    assert (emb.cpu() - ref).abs().max().item() < EMB_TOL


def test_c3_distance_stage_full_size(engine):
    """Config C3's distance stage at full size: 10 000 degraded x 1 000 references (unit-norm embeddings)."""
    gen = torch.Generator().manual_seed(3)
    deg = torch.nn.functional.normalize(torch.randn(10000, 256, generator=gen), dim=1)
    ref = torch.nn.functional.normalize(torch.randn(1000, 256, generator=gen), dim=1)
    ref[7] = deg[123]
    d, m = engine.pairwise(deg.cuda(), ref.cuda())
    torch.cuda.synchronize()
    assert d.shape == (10000, 1000) and d.dtype == torch.float64
    assert d[123, 7].item() == 0.0
    rows = [0, 123, 5000, 9999]
    dref, mref = O.pairwise(deg[rows].numpy(), ref.numpy())
    assert np.abs(d[rows].cpu().numpy() - dref).max() < 1e-13
    assert np.abs(m[rows].cpu().numpy() - mref).max() < 1e-13
    assert (m - d.mean(dim=1)).abs().max().item() < 1e-12   # mean of means identity over the whole matrix


def test_ragged_batch_is_bit_identical_to_per_clip(engine):
    """Clips of different lengths in one launch sequence == each clip on its own (the reference's per-file loop)."""
    gen = torch.Generator().manual_seed(21)
    lens = [16384, 400, 27225, 9001, 64000, 30267, 5000, 12345]      # includes the conv stack's minimum (T = 1)
    waves = [(0.1 * torch.randn(n, generator=gen)).clamp(-1, 1) for n in lens]
    emb = engine.embed_ragged(waves)
    torch.cuda.synchronize()
    assert emb.shape == (len(lens), 256) and torch.isfinite(emb).all()
    for i, w in enumerate(waves):
        one = engine.embed(w[None, :].cuda())
        assert torch.equal(one[0], emb[i]), (i, lens[i], (one[0] - emb[i]).abs().max().item())
    # a ragged batch of equal lengths equals the uniform path too
    same = [waves[0], waves[0].flip(0)]
    assert torch.equal(engine.embed_ragged(same), engine.embed(torch.stack(same).cuda()))


def test_ragged_example_wavs_vs_hf_golden(engine):
    g = np.load(os.path.join(GOLD, "hf_example_wavs.npz"))
    waves, want = [], []
    for d, names, embs in (("nmr-data", g["nmr_names"], g["nmr_emb"]), ("test-data", g["deg_names"], g["deg_emb"])):
        for n, e in zip(names, embs):
            waves.append(O.load_processing(os.path.join(GOLD, "wavs", d, f"{n}.wav"))[0])
            want.append(e)
    emb = engine.embed_ragged(waves).cpu().numpy()       # six clips, six different lengths, one launch sequence
    assert np.abs(emb - np.stack(want)).max() < EMB_TOL


def test_predict_csv_mode_and_cli(built_lib, sd0, tmp_path, monkeypatch):
    """csv mode (nomad.py:91-95,154-159): a `filename` column of wav paths; same scores as dir mode; CLI entry."""
    import pandas as pd
    from nomad_amd.nomad import Nomad
    g = np.load(os.path.join(GOLD, "hf_example_wavs.npz"))
    nmr_csv, deg_csv = tmp_path / "nmr.csv", tmp_path / "deg.csv"
    pd.DataFrame({"filename": [os.path.join(GOLD, "wavs", "nmr-data", f"{n}.wav") for n in g["nmr_names"]]}).to_csv(nmr_csv, index=False)
    pd.DataFrame({"filename": [os.path.join(GOLD, "wavs", "test-data", f"{n}.wav") for n in g["deg_names"]]}).to_csv(deg_csv, index=False)
    nmd = Nomad(weights=sd0)
    avg, dm = nmd.predict("csv", str(nmr_csv), str(deg_csv), results_path=str(tmp_path))
    assert list(dm.columns) == [str(n) for n in g["nmr_names"]]          # csv order is preserved
    assert np.abs(dm.to_numpy() - np.round(g["dist"], 3)).max() <= 1.01e-3
    assert np.abs(avg["NOMAD"].to_numpy() - np.round(g["mean"], 3)).max() <= 1.01e-3
    bad = tmp_path / "bad.csv"
    pd.DataFrame({"path": ["x.wav"]}).to_csv(bad, index=False)
    with pytest.raises(Exception, match="column called filename"):
        nmd.get_embeddings(str(bad))
    # results_path=None writes results-csv/<dd-mm-YYYY_HH-MM-SS>/..._nomad_avg.csv (nomad.py:122-133)
    monkeypatch.chdir(tmp_path)
    nmd.predict("csv", str(nmr_csv), str(deg_csv))
    stamped = os.listdir(tmp_path / "results-csv")
    assert len(stamped) == 1
    files = sorted(os.listdir(tmp_path / "results-csv" / stamped[0]))
    assert files[0].endswith("_nomad_avg.csv") and files[1].endswith("_nomad_scores.csv")


def test_plain_c_client_of_the_abi(built_lib, tmp_path):
    """The boundary is a C ABI: a C99 program (no Python, no torch, no C++) drives create / embed / pairwise."""
    import shutil
    import subprocess
    from conftest import ROOT
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.join(ROOT, "nomad_amd")
    cmd = [gcc, "-std=c99", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-L" + libdir, "-lnomad_hip", "-L/opt/rocm/lib", "-lamdhip64",
           "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "status = 0" in res.stdout


@pytest.mark.parametrize("kind", ["silence", "dc_offset", "square", "impulse", "loud_noise"])
def test_degenerate_waveforms_vs_oracle(engine, sd0, kind):
    """Inputs that stress the GroupNorm statistics (zero variance, large mean^2 / variance) and saturation."""
    gen = torch.Generator().manual_seed(17)
    n = 24000
    t = torch.arange(n, dtype=torch.float32)
    wav = {
        "silence": torch.zeros(n),
        "dc_offset": 0.5 + 1e-3 * torch.randn(n, generator=gen),
        "square": torch.sign(torch.sin(2 * torch.pi * 220.0 * t / 16000.0)),
        "impulse": torch.zeros(n).index_fill_(0, torch.tensor([n // 2]), 1.0),
        "loud_noise": torch.randn(n, generator=gen).clamp(-1, 1),
    }[kind][None, :]
    with torch.no_grad():
        ref = O.triplet_forward(sd0, wav)
    emb = engine.embed(wav.cuda()).cpu()
    assert torch.isfinite(emb).all()
    # dc_offset: mean^2 / variance of the conv output is ~1e5, where torch's fp32 group_norm itself carries ~1e-4
    tol = 2e-4 if kind == "dc_offset" else 2e-5
    assert (emb - ref).abs().max().item() < tol, (kind, (emb - ref).abs().max().item())


def test_repeat_runs_are_bit_identical(engine):
    """Race screen for the multi-stage LDS-DMA pipelines: the same batch must give the same bits every time, also
    while another stream keeps the chip busy (a hand-off that only works on an idle GPU shows up here)."""
    gen = torch.Generator().manual_seed(33)
    wav = (0.1 * torch.randn(64, 64000, generator=gen)).clamp(-1, 1).cuda()
    ref = engine.embed(wav).clone()
    ref16 = engine.embed_bf16(wav).clone()
    refx3 = engine.embed_bf16x3(wav).clone()
    side = torch.cuda.Stream()
    junk = torch.randn(4096, 4096, device="cuda")
    for it in range(12):
        with torch.cuda.stream(side):
            for _ in range(4):
                junk = junk @ junk * 1e-3          # uneven background load on the other queue
        assert torch.equal(engine.embed(wav), ref), it
        if it % 4 == 2:
            outx3 = engine.embed_bf16x3(wav)           # two halves on two streams (Engine.X3_SPLIT_ROWS)
            if not torch.equal(outx3, refx3):
                rows = torch.nonzero((outx3 != refx3).any(dim=1)).flatten().tolist()
                pytest.fail(f"iteration {it}: embed_bf16x3 differs from its first result in clips {rows} "
                            f"(max|diff| {(outx3 - refx3).abs().max().item():.3e})")
        if it % 4 == 0:
            out16 = engine.embed_bf16(wav)
            if not torch.equal(out16, ref16):     # say what differs before failing: which clips, by how much, and whether it repeats
                rows = torch.nonzero((out16 != ref16).any(dim=1)).flatten().tolist()
                again = engine.embed_bf16(wav)
                torch.cuda.synchronize()
                stale = [r for r in rows if torch.equal(out16[r], ref[r])]      # rows that hold the fp32 result of the call before?
                pytest.fail(f"iteration {it}: embed_bf16 differs from its first result in clips {rows} "
                            f"[of these equal to the fp32 embedding bit for bit: {stale}] "
                            f"(max|diff| {(out16 - ref16).abs().max().item():.3e}, finite {bool(torch.isfinite(out16).all())}); "
                            f"the next call {'equals' if torch.equal(again, ref16) else 'differs from'} the first result"
                            f"{'' if torch.equal(again, ref16) else ' in clips ' + str(torch.nonzero((again != ref16).any(dim=1)).flatten().tolist())}")
    torch.cuda.synchronize()


def test_large_batch_crosses_2_pow_31_elements(engine):
    """512 x 4 s clips: the conv0 output alone is 3.36e9 floats (> 2^31, > 2^32 bytes several times over), so any 32-bit
    element or byte offset anywhere in the path would corrupt the late clips.  Batch invariance is bit-exact, so the
    first and last clips must equal what a batch of two gives."""
    g = torch.Generator().manual_seed(77)
    wav = (0.1 * torch.randn(512, 64000, generator=g)).clamp(-1, 1).cuda()
    emb = engine.embed(wav)
    ref = engine.embed(torch.stack([wav[0], wav[511]]).contiguous())
    mid = engine.embed(torch.stack([wav[300], wav[421]]).contiguous())
    assert torch.equal(emb[0], ref[0]) and torch.equal(emb[511], ref[1])
    assert torch.equal(emb[300], mid[0]) and torch.equal(emb[421], mid[1])
    e16 = engine.embed_bf16(wav)
    r16 = engine.embed_bf16(torch.stack([wav[0], wav[511]]).contiguous())
    assert torch.equal(e16[0], r16[0]) and torch.equal(e16[511], r16[1])
    del wav, emb, e16
    engine._ws = None  # hand the 26 GB workspace back
    torch.cuda.empty_cache()
