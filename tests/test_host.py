"""CPU: host-side logic - checkpoint layout, WAV front end, partitioning."""
import math
import os

import numpy as np
import pytest
import torch

from conftest import GOLD
from nomad_amd import wavio
from nomad_amd.weights import check_state_dict, expected_shapes, load_checkpoint, seeded_state_dict


def test_state_dict_layout(sd0, tmp_path):
    shapes = expected_shapes()
    assert len(shapes) == 1 + 7 + 2 + 2 + 2 + 3 + 12 * 16 + 2 + 2
    n_params = sum(int(np.prod(s)) for k, s in shapes.items())
    assert 94.3e6 < n_params < 94.8e6  # wav2vec2-base + head (~94.6 M)
    check_state_dict(sd0)
    bad = dict(sd0)
    del bad["embedding_layer.1.bias"]
    with pytest.raises(KeyError):
        check_state_dict(bad)
    # round trip through the on-disk format the reference uses (torch.save of a state dict)
    p = tmp_path / "nomad_best_model.pt"
    small = {k: v for k, v in sd0.items()}
    torch.save(small, p)
    back = load_checkpoint(str(p))
    assert all(torch.equal(back[k], sd0[k]) for k in shapes if not k.endswith("mask_emb"))


def test_seeded_weights_are_deterministic():
    a, b = seeded_state_dict(3), seeded_state_dict(3)
    assert all(torch.equal(a[k], b[k]) for k in a)
    c = seeded_state_dict(4)
    assert not torch.equal(a["embedding_layer.1.weight"], c["embedding_layer.1.weight"])


def test_wav_reader_formats(tmp_path):
    from scipy.io import wavfile
    rng = np.random.default_rng(0)
    x16 = (rng.standard_normal((800, 2)) * 8000).astype(np.int16)
    wavfile.write(tmp_path / "s16.wav", 22050, x16)
    y, sr = wavio.read_wav(str(tmp_path / "s16.wav"))
    assert sr == 22050 and y.shape == (2, 800)
    assert np.array_equal(y, x16.T.astype(np.float32) / 32768.0)
    xf = rng.standard_normal(500).astype(np.float32) * 0.1
    wavfile.write(tmp_path / "f32.wav", 16000, xf)
    y, sr = wavio.read_wav(str(tmp_path / "f32.wav"))
    assert np.array_equal(y[0], xf)
    x32 = (rng.standard_normal(300) * 1e8).astype(np.int32)
    wavfile.write(tmp_path / "s32.wav", 8000, x32)
    y, _ = wavio.read_wav(str(tmp_path / "s32.wav"))
    assert np.allclose(y[0], x32 / 2147483648.0, atol=1e-7)
    with pytest.raises(ValueError):
        (tmp_path / "bad.wav").write_bytes(b"not a wav file at all")
        wavio.read_wav(str(tmp_path / "bad.wav"))


def test_load_processing_matches_reference_semantics(tmp_path):
    from scipy.io import wavfile
    # shipped fixture: 16 kHz mono PCM16 -> (1, N) fp32, untouched
    w = wavio.load_processing(os.path.join(GOLD, "wavs", "nmr-data", "FI53_04.wav"))
    assert w.shape == (1, 30671) and w.dtype == np.float32 and np.abs(w).max() <= 1.0
    # stereo: (ch0 + ch1) / 2 (nomad.py:199-200); trim to 10 s (nomad.py:208-210)
    rng = np.random.default_rng(1)
    st = (rng.standard_normal((16000 * 11, 2)) * 3000).astype(np.int16)
    wavfile.write(tmp_path / "st.wav", 16000, st)
    w = wavio.load_processing(str(tmp_path / "st.wav"), trim=True)
    assert w.shape == (1, 160000)
    ref = (st[:160000, 0].astype(np.float32) / 32768.0 + st[:160000, 1].astype(np.float32) / 32768.0) / 2
    assert np.allclose(w[0], ref, atol=1e-7)


@pytest.mark.parametrize("sr", [8000, 22050, 44100, 48000])
def test_resampler_properties(sr):
    """torchaudio is not installable here, so the sinc resampler is pinned by its defining properties:
    output length ceil(N*16000/sr), a band-limited tone is reproduced, DC gain is 1."""
    n = sr  # one second
    t = np.arange(n) / sr
    tone = (0.5 * np.sin(2 * np.pi * 440.0 * t)).astype(np.float32)[None, :]
    y = wavio.resample(tone, sr, 16000)
    assert y.shape == (1, math.ceil(n * 16000 / sr))
    t2 = np.arange(y.shape[1]) / 16000
    want = 0.5 * np.sin(2 * np.pi * 440.0 * t2)
    mid = slice(200, -200)
    assert np.abs(y[0, mid] - want[mid]).max() < 2e-3
    dc = wavio.resample(np.ones((1, n), dtype=np.float32), sr, 16000)
    assert np.abs(dc[0, mid] - 1.0).max() < 2e-3


def test_partition_covers_everything():
    from nomad_amd.dist import partition
    for n in (0, 1, 7, 1000, 10000):
        for w in (1, 2, 4, 8):
            spans = [partition(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


# ---- nomad_amd.train host logic (no GPU) -----------------------------------------------------------------------------
def test_triplet_dataset_collate_and_schedule(tmp_path):
    import struct
    import numpy as np
    import pandas as pd
    import torch
    from nomad_amd.train import ExponentialLR, TripletDataset

    def write(path, n, sr=16000):
        pcm = (np.clip(0.1 * np.random.RandomState(n).randn(n), -1, 1) * 32767).astype("<i2").tobytes()
        with open(path, "wb") as f:
            f.write(b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVEfmt " +
                    struct.pack("<IHHIIHH", 16, 1, 1, sr, sr * 2, 2, 16) + b"data" + struct.pack("<I", len(pcm)) + pcm)

    rows = []
    for i, n in enumerate((3000, 170000, 5000)):  # the middle one is longer than 10 s
        for role in ("Anchor", "Positive", "Negative"):
            write(str(tmp_path / f"{role}_{i}.wav"), n + 10 * len(role))
        rows.append(dict(Anchor=f"/Anchor_{i}.wav", Positive=f"/Positive_{i}.wav", Negative=f"/Negative_{i}.wav", db=1 + (i == 2)))
    rows.append(dict(rows[0]))  # duplicate row: dropped (triplet_dataloader.py:46)
    csv = str(tmp_path / "t.csv")
    pd.DataFrame(rows).to_csv(csv, index=False)
    cfg = dict(root=str(tmp_path), train_df=csv, trim=True)
    ds = TripletDataset(cfg, "train_df", level=[1, 2])
    assert len(ds) == 3
    assert len(TripletDataset(cfg, "train_df", level=[2])) == 1
    a, p, n = ds[1]
    assert a.shape == (1, 160000) and p.shape == (1, 160000)  # trimmed to 10 s
    A, P, N = ds.collate_fn([ds[0], ds[2]])
    assert A.shape == (2, 1, 5000 + 60) and A.dtype == torch.float32
    assert torch.all(A[0, 0, 3060:] == 0) and A[0, 0, :3060].abs().sum() > 0  # zero padded to the batch maximum
    cfg["trim"] = False
    assert TripletDataset(cfg, "train_df")[1][0].shape == (1, 170060)
    sch = ExponentialLR([1e-5, 1e-4], 0.99)
    sch.step(); sch.step()
    ref = torch.optim.lr_scheduler.ExponentialLR(torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1e-4), gamma=0.99)
    ref.optimizer.step(); ref.step(); ref.step()
    assert abs(sch.get_last_lr()[1] - ref.get_last_lr()[0]) < 1e-12 and abs(sch.get_last_lr()[0] - 1e-5 * 0.99 ** 2) < 1e-15


def test_load_pretrained_accepts_a_fairseq_style_checkpoint(tmp_path):
    import torch
    from nomad_amd.train import load_pretrained
    from nomad_amd.weights import check_state_dict, seeded_state_dict
    sd = seeded_state_dict(4)
    model = {k[len("ssl_model."):]: v for k, v in sd.items() if k.startswith("ssl_model.")}
    model["quantizer.vars"] = torch.zeros(3)          # pre-training modules the reference removes
    model["final_proj.weight"] = torch.zeros(2, 2)
    path = str(tmp_path / "wav2vec_small.pt")
    torch.save({"model": model, "cfg": {"whatever": 1}}, path)
    got = load_pretrained(path)
    check_state_dict(got)
    assert all(torch.equal(got[k], sd[k]) for k in sd if k.startswith("ssl_model."))
    w = got["embedding_layer.1.weight"]
    assert w.shape == (256, 768) and w.abs().max() <= 1 / 768 ** 0.5 and w.std() > 0.01
    nomad_path = str(tmp_path / "nomad.pt")
    torch.save(sd, nomad_path)
    assert torch.equal(load_pretrained(nomad_path)["embedding_layer.1.weight"], sd["embedding_layer.1.weight"])


def test_load_pretrained_never_falls_back_to_the_full_unpickler_on_its_own(tmp_path, monkeypatch):
    """a file the restricted loader rejects is exactly the file that must not reach pickle's full machinery
    unless the caller says so."""
    import pickle
    import torch
    from nomad_amd.train import load_pretrained
    from nomad_amd.weights import seeded_state_dict
    marker = tmp_path / "executed"

    class Evil:
        def __reduce__(self):
            return (open, (str(marker), "w"))
    path = str(tmp_path / "evil.pt")
    sd = seeded_state_dict(4)
    model = {k[len("ssl_model."):]: v for k, v in sd.items() if k.startswith("ssl_model.")}
    torch.save({"model": model, "cfg": Evil()}, path, pickle_protocol=pickle.DEFAULT_PROTOCOL)
    monkeypatch.delenv("NOMAD_ALLOW_UNSAFE_PICKLE", raising=False)
    with pytest.raises(RuntimeError, match="allow_unsafe_pickle"):
        load_pretrained(path)
    assert not marker.exists()
    got = load_pretrained(path, allow_unsafe_pickle=True)      # the caller's explicit decision
    assert marker.exists() and "ssl_model.layer_norm.weight" in got


def test_feature_grad_mult_is_never_read_from_an_unnamed_file(tmp_path, monkeypatch):
    """Nomad() construction must not unpickle ./pt-models/wav2vec_small.pt behind the caller's back: the
    constant 0.1 unless a checkpoint is NAMED, and then only through the restricted unpickler."""
    import torch
    from nomad_amd import weights as W
    (tmp_path / "pt-models").mkdir()
    torch.save({"cfg": {"model": {"feature_grad_mult": 0.5}}}, str(tmp_path / "pt-models" / "wav2vec_small.pt"))
    monkeypatch.chdir(tmp_path)
    monkeypatch.delenv("NOMAD_W2V_CHECKPOINT", raising=False)
    calls = []
    real_load = torch.load

    def spy(*a, **k):
        calls.append(k.get("weights_only"))
        return real_load(*a, **k)
    monkeypatch.setattr(torch, "load", spy)
    assert W.find_feature_grad_mult() == 0.1 and calls == []
    named = str(tmp_path / "named.pt")
    torch.save({"cfg": {"model": {"feature_grad_mult": 0.25}}}, named)
    assert W.find_feature_grad_mult(named) == 0.25
    monkeypatch.setenv("NOMAD_W2V_CHECKPOINT", named)
    assert W.find_feature_grad_mult() == 0.25
    assert calls == [True, True]                      # never the full unpickler

    class Evil:                                       # a pickle that would run code: refused, constant returned
        def __reduce__(self):
            return (os.system, ("true",))
    evil = str(tmp_path / "evil.pt")
    torch.save({"cfg": Evil()}, evil)
    assert W.find_feature_grad_mult(evil) == 0.1


# ---- the decode -> pack -> launch pipeline of Nomad.get_embeddings_csv (host logic, fake engine) ----------------------
class _FakeEngine:
    """Records what the pipeline asks of an Engine; "embeds" a clip as [length, first sample, ...]."""

    def __init__(self):
        self.alive = 0          # packed batches not yet dropped by the consumer
        self.max_alive = 0
        self.batches = []

    def pack_ragged_host(self, waves):
        import weakref
        self.alive += 1
        self.max_alive = max(self.max_alive, self.alive)
        lens = [int(w.shape[0]) for w in waves]
        host = np.zeros((len(waves), max(lens)), dtype=np.float32)
        for i, w in enumerate(waves):
            host[i, :lens[i]] = w

        class Buf:                       # something that can carry a finalizer (a stand-in for the pinned tensor)
            pass
        b = Buf()
        b.host = host
        weakref.finalize(b, self._dropped)
        return b, lens

    def _dropped(self):
        self.alive -= 1

    def embed_ragged(self, waves, precision=None, packed=None):
        assert waves is None
        b, lens = packed
        self.batches.append(list(lens))
        out = np.zeros((len(lens), 256), dtype=np.float32)
        out[:, 0] = lens
        out[:, 1] = b.host[:, 0]
        return out

    def fetch_async(self, emb):
        class F:
            def result(self_inner):
                return emb
        return F()


def _fake_nomad(eng, loads):
    from nomad_amd.nomad import Nomad
    n = Nomad.__new__(Nomad)            # no GPU: only the host pipeline is under test
    n.engine, n.precision = eng, "fp32"
    n.load_processing = lambda p, trim=False: loads(p)
    return n


def test_embedding_pipeline_order_batching_and_bounded_memory():
    import pandas as pd
    rng = np.random.default_rng(0)
    lens = [int(x) for x in rng.integers(100, 4000, size=57)] + [9000] + [50, 60]   # one file longer than a whole batch
    paths = [f"f{i}.wav" for i in range(len(lens))]

    def load(p):
        i = int(p[1:-4])
        return np.full((1, lens[i]), float(i), dtype=np.float32)
    eng = _FakeEngine()
    n = _fake_nomad(eng, load)
    df = n.get_embeddings_csv(None, pd.DataFrame({"filename": paths}), max_batch_samples=8000)
    assert list(df["filename"]) == paths                                  # listing order kept
    assert [int(x) for x in df[0]] == lens and [int(x) for x in df[1]] == list(range(len(lens)))
    assert [l for b in eng.batches for l in b] == lens                    # consecutive files, every file exactly once
    assert all(sum(b) <= 8000 or len(b) == 1 for b in eng.batches) and [9000] in eng.batches
    assert len(eng.batches) > 5
    assert eng.max_alive <= 2 + 1     # PIPELINE_BATCHES staged batches (+1 for the instant the consumer swaps batches)


def test_embedding_pipeline_propagates_decode_errors_and_handles_empty():
    import pandas as pd

    def load(p):
        if p == "bad.wav":
            raise ValueError("bad.wav: not a RIFF/WAVE file")
        return np.zeros((1, 500), dtype=np.float32)
    n = _fake_nomad(_FakeEngine(), load)
    with pytest.raises(ValueError, match="bad.wav"):
        n.get_embeddings_csv(None, pd.DataFrame({"filename": ["a.wav"] * 40 + ["bad.wav"] + ["b.wav"] * 40}), max_batch_samples=2000)
    df = n.get_embeddings_csv(None, pd.DataFrame({"filename": []}))
    assert df.shape[0] == 0


# ---- resampler parity: the product's form vs the separately written restatement of torchaudio's formulation -----------
@pytest.mark.parametrize("sr", [8000, 22050, 44100, 48000, 11025, 32000])
def test_resampler_matches_the_torchaudio_restatement(sr):
    """nomad.py:203-205: torchaudio.transforms.Resample(sr, 16000).  wavio.resample (numpy, strided windows + einsum)
    against oracle/resample_oracle.py (torch, per-phase kernels + conv1d(stride=orig), torchaudio's own formulation):
    same length, samples within 1e-6 (fp32 summation order is the only difference)."""
    import torch
    from oracle import resample_oracle as R
    rng = np.random.default_rng(sr)
    x = (0.3 * rng.standard_normal((2, sr + 123))).astype(np.float32)       # 1 s + an awkward tail, two channels
    y = wavio.resample(x, sr, 16000)
    ref = R.resample(torch.from_numpy(x), sr, 16000).numpy()
    assert y.shape == ref.shape == (2, math.ceil((sr + 123) * 16000 / sr))
    assert np.abs(y - ref).max() < 1e-6, np.abs(y - ref).max()
    # kernel itself: identical taps
    k, w, orig, new = wavio._sinc_kernel(sr, 16000)
    kref, wref = R.sinc_resample_kernel(orig, new)
    assert w == wref and k.shape == tuple(kref.shape[::2]) and np.abs(k - kref[:, 0].numpy()).max() < 1e-7


def test_load_processing_resamples_like_the_restatement(tmp_path):
    """End of the front end: a 44.1 kHz stereo PCM-16 file through load_processing == mono mix then the restated
    resampler (nomad.py:199-205 order: mix first, resample second)."""
    import torch
    from scipy.io import wavfile
    from oracle import resample_oracle as R
    rng = np.random.default_rng(7)
    st = (rng.standard_normal((44100, 2)) * 4000).astype(np.int16)
    wavfile.write(tmp_path / "s.wav", 44100, st)
    w = wavio.load_processing(str(tmp_path / "s.wav"))
    mono = (st[:, 0].astype(np.float32) / 32768.0 + st[:, 1].astype(np.float32) / 32768.0) / 2
    ref = R.resample(torch.from_numpy(mono[None, :]), 44100, 16000).numpy()
    assert w.shape == ref.shape == (1, 16000) and np.abs(w - ref).max() < 1e-6


def test_fast_scores_csv_is_byte_identical_to_pandas(tmp_path):
    """Nomad.predict writes the N_deg x N_ref table with a lookup-table formatter (pandas needs seconds for 10^7 cells):
    the bytes must equal DataFrame.to_csv's, and anything off the 3-decimal grid must take the pandas path."""
    import pandas as pd
    from nomad_amd.nomad import _write_rounded_csv
    rng = np.random.default_rng(0)
    vals = np.round(rng.uniform(0, 2, (300, 40)), 3)
    vals[0, :8] = [0.0, 1.0, 2.0, 0.001, 0.01, 0.1, 1.999, 0.5]
    for case, v, labels in (("grid", vals, [f"clip_{i:04d}" for i in range(300)]),
                            ("offgrid", vals + 1e-4, [f"clip_{i:04d}" for i in range(300)]),
                            ("nan", np.where(rng.uniform(size=vals.shape) < 0.01, np.nan, vals), [f"c{i}" for i in range(300)]),
                            ("quoting", vals, [f'we,ird "{i}"' for i in range(300)]),
                            ("float32", vals.astype(np.float32).astype(np.float64).round(3), [f"c{i}" for i in range(300)])):
        df = pd.DataFrame(v)
        df["Test File"] = labels
        df.set_index("Test File", inplace=True)
        df.columns = [f"ref_{j}" for j in range(v.shape[1])]
        a, b = tmp_path / f"{case}_pandas.csv", tmp_path / f"{case}_fast.csv"
        df.reset_index().to_csv(a, index=False)
        _write_rounded_csv(df.reset_index(), str(b))
        assert a.read_bytes() == b.read_bytes(), case
    back = pd.read_csv(tmp_path / "grid_fast.csv", index_col="Test File")
    assert np.array_equal(back.to_numpy(), vals)
