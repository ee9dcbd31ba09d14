"""The C ABI's WAV reader (nomad_wav_probe / nomad_wav_read_rows, host only) against the Python front end it shortcuts
(nomad_amd/wavio.py: read_wav + the two-channel mean of load_processing, /root/reference/src/nomad_audio/nomad.py:196-200),
and the file pipeline of Nomad.get_embeddings_csv with the native reader on real files (fake engine: no GPU here)."""
import os
import struct

import numpy as np
import pandas as pd
import pytest

from nomad_amd import _lib, wavio


def _wav_bytes(samples: np.ndarray, sr: int, tag: int, bits: int, extensible=False, junk=b"", cut=0) -> bytes:
    """samples: (frames, channels) float64 in [-1, 1).  junk: an extra chunk (odd sizes exercise the pad byte) before 'data'."""
    ch = samples.shape[1]
    if tag == 1 and bits == 8:
        pcm = np.clip(np.round(samples * 128 + 128), 0, 255).astype(np.uint8).tobytes()
    elif tag == 1 and bits == 16:
        pcm = np.clip(np.round(samples * 32768), -32768, 32767).astype("<i2").tobytes()
    elif tag == 1 and bits == 24:
        v = np.clip(np.round(samples * 8388608), -8388608, 8388607).astype(np.int64).reshape(-1)
        pcm = b"".join(int(x & 0xFFFFFF).to_bytes(3, "little") for x in v)
    elif tag == 1 and bits == 32:
        pcm = np.clip(np.round(samples * 2147483648), -2147483648, 2147483647).astype("<i4").tobytes()
    elif tag == 3 and bits == 32:
        pcm = samples.astype("<f4").tobytes()
    else:
        pcm = samples.astype("<f8").tobytes()
    block = ch * bits // 8
    if extensible:
        guid = struct.pack("<H", tag) + b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71"
        fmt = struct.pack("<HHIIHH", 0xFFFE, ch, sr, sr * block, block, bits) + struct.pack("<HHI", 22, bits, 3) + guid
    else:
        fmt = struct.pack("<HHIIHH", tag, ch, sr, sr * block, block, bits)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt
    if junk:
        body += b"LIST" + struct.pack("<I", len(junk)) + junk + (b"\x00" if len(junk) & 1 else b"")
    body += b"data" + struct.pack("<I", len(pcm)) + pcm
    data = b"RIFF" + struct.pack("<I", len(body)) + body
    return data[:len(data) - cut] if cut else data


CASES = [  # (name, tag, bits, channels, extensible, junk, cut bytes)
    ("pcm16", 1, 16, 1, False, b"", 0), ("pcm16_stereo", 1, 16, 2, False, b"", 0), ("pcm16_3ch", 1, 16, 3, False, b"", 0),
    ("pcm8", 1, 8, 1, False, b"", 0), ("pcm24", 1, 24, 1, False, b"", 0), ("pcm24_stereo", 1, 24, 2, False, b"", 0),
    ("pcm32", 1, 32, 2, False, b"", 0), ("f32", 3, 32, 1, False, b"", 0), ("f64_stereo", 3, 64, 2, False, b"", 0),
    ("ext_pcm16", 1, 16, 1, True, b"", 0), ("ext_f32_stereo", 3, 32, 2, True, b"", 0),
    ("junk_odd", 1, 16, 1, False, b"INFOabc", 0), ("junk_even", 1, 16, 2, False, b"INFOabcd", 0),
    ("cut_tail", 1, 16, 1, False, b"", 400), ("cut_tail_24", 1, 24, 2, False, b"", 301),
    ("long", 1, 16, 1, False, b"", 0),
]


@pytest.fixture(scope="module")
def wav_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("wavs")
    rng = np.random.default_rng(7)
    paths = []
    for k, (name, tag, bits, ch, ext, junk, cut) in enumerate(CASES):
        frames = 200_000 if name == "long" else int(rng.integers(900, 5000))   # "long": more than one conversion block
        x = np.clip(0.4 * rng.standard_normal((frames, ch)), -0.999, 0.999)
        p = str(d / f"{k:02d}_{name}.wav")
        with open(p, "wb") as f:
            f.write(_wav_bytes(x, 16000, tag, bits, ext, junk, cut))
        paths.append(p)
    return paths


def test_native_reader_is_bit_identical_to_the_python_front_end(wav_dir):
    info, status = wavio.probe(wav_dir, threads=3)
    assert status == [0] * len(wav_dir)
    want = [wavio.load_processing(p)[0] for p in wav_dir]
    assert [int(i.frames) for i in info] == [w.shape[0] for w in want]
    assert all(i.sample_rate == 16000 for i in info)
    stride = max(w.shape[0] for w in want) + 5
    for threads in (1, 4):
        host = np.full((len(wav_dir) + 2, stride), np.float32(7.0))
        rows = list(range(len(wav_dir), 0, -1))                    # any row order; row 0 and the last row stay untouched
        wavio.read_rows(wav_dir, list(info), rows, host, threads=threads)
        for r, w in zip(rows, want):
            assert np.array_equal(host[r, :w.shape[0]].view(np.uint32), w.view(np.uint32))
            assert np.all(host[r, w.shape[0]:] == 7.0)              # nothing written past the clip
        assert np.all(host[0] == 7.0) and np.all(host[-1] == 7.0)


def test_native_reader_takes_torch_tensors_and_checks_its_arguments(wav_dir):
    import torch
    info, _ = wavio.probe(wav_dir[:2])
    host = torch.zeros(2, int(max(info[0].frames, info[1].frames)))
    wavio.read_rows(wav_dir[:2], [info[0], info[1]], [0, 1], host)
    assert wavio.frames_at(info[0]) == info[0].frames and wavio.frames_at(info[0], 8000) == -(-info[0].frames // 2)
    assert np.array_equal(host[0, :info[0].frames].numpy(), wavio.load_processing(wav_dir[0])[0])
    with pytest.raises(_lib.NomadHipError, match="do not fit"):
        wavio.read_rows(wav_dir[:1], [info[0]], [0], torch.zeros(1, 10))


def test_probe_reports_what_it_does_not_decode(tmp_path):
    x = np.zeros((800, 1))
    files = {
        "ok8k.wav": _wav_bytes(x, 8000, 1, 16),                      # decodable; read_rows resamples it
        "alaw.wav": _wav_bytes(x, 16000, 1, 16).replace(struct.pack("<HH", 1, 1), struct.pack("<HH", 6, 1), 1),
        "ragged_tail.wav": _wav_bytes(x, 16000, 1, 16, cut=1),       # np.frombuffer rejects it: so does the probe
        "nodata.wav": _wav_bytes(x, 16000, 1, 16)[:36],
        "text.wav": b"hello, this is not audio",
        "empty.wav": b"",
    }
    paths = []
    for name, data in files.items():
        p = str(tmp_path / name)
        open(p, "wb").write(data)
        paths.append(p)
    paths.append(str(tmp_path / "missing.wav"))
    os.mkdir(tmp_path / "dir.wav")
    paths.append(str(tmp_path / "dir.wav"))
    info, status = wavio.probe(paths, threads=2)
    assert status[0] == 0 and info[0].sample_rate == 8000 and info[0].frames == 800
    assert status[1:6] == [-6] * 5                                    # NOMAD_ERR_FORMAT
    assert status[6:] == [-5, -5]                                     # NOMAD_ERR_IO
    for p in paths[1:6]:                                              # ... and the Python front end refuses the same files
        with pytest.raises((ValueError, struct.error)):
            wavio.read_wav(p)
    assert wavio.probe([], 4)[1] == []


# ---- the file pipeline on real files -------------------------------------------------------------------------------
class _Engine:
    """Stands in for nomad_amd.engine.Engine: the "embedding" of a clip is (length, sum, first, last sample)."""

    def __init__(self):
        self.batches, self.kinds = [], []

    def pack_ragged_host(self, waves):
        lens = [int(w.shape[0]) for w in waves]
        host = np.zeros((len(waves), max(lens)), dtype=np.float32)
        for i, w in enumerate(waves):
            host[i, :lens[i]] = np.asarray(w)
        self.kinds.append("python")
        return host, lens

    def embed_ragged(self, waves, precision=None, packed=None):
        host, lens = packed
        if not isinstance(host, np.ndarray):
            self.kinds.append("native")
            host = host.numpy()
        self.batches.append(list(lens))
        out = np.zeros((len(lens), 256), dtype=np.float32)
        for i, n in enumerate(lens):
            out[i, :4] = n, host[i, :n].astype(np.float64).sum(), host[i, 0], host[i, n - 1]
        return out

    def fetch_async(self, emb):
        class F:
            def result(self_inner):
                return emb
        return F()


def _nomad(eng, native_threads):
    import torch
    from nomad_amd.nomad import Nomad
    n = Nomad.__new__(Nomad)
    n.engine, n.precision = eng, "fp32"
    n.NATIVE_WAV_THREADS = native_threads
    n.load_processing = lambda p, trim=False: torch.from_numpy(wavio.load_processing(p, 16000, trim))
    return n


def test_file_pipeline_native_equals_python(tmp_path):
    rng = np.random.default_rng(3)
    paths = []
    for i in range(90):
        sr = 8000 if i % 7 == 3 else 16000                         # every 7th file needs the resampler (native too)
        ch = 2 if i % 5 == 0 else 1
        bits = (16, 24, 32)[i % 3]
        x = np.clip(0.3 * rng.standard_normal((int(rng.integers(400, 6000)), ch)), -0.99, 0.99)
        p = str(tmp_path / f"{i:03d}.wav")
        open(p, "wb").write(_wav_bytes(x, sr, 1, bits))
        paths.append(p)
    df = pd.DataFrame({"filename": paths})
    e_py, e_nat = _Engine(), _Engine()
    ref = _nomad(e_py, 0).get_embeddings_csv(None, df, max_batch_samples=30_000)
    got = _nomad(e_nat, 3).get_embeddings_csv(None, df, max_batch_samples=30_000)
    assert e_py.batches == e_nat.batches and len(e_py.batches) > 5      # same batching either way (resampled lengths too)
    assert set(e_py.kinds) == {"python"} and set(e_nat.kinds) == {"native"}
    a, b = ref[list(range(256))].to_numpy(), got[list(range(256))].to_numpy()
    same_rate = np.array([i % 7 != 3 for i in range(90)])
    assert np.array_equal(a[same_rate].view(np.uint32), b[same_rate].view(np.uint32))      # decode only: bit-identical
    assert np.array_equal(a[:, 0], b[:, 0]) and np.allclose(a, b, rtol=0, atol=2e-5)        # resampled: fp32 summation order
    assert list(got["filename"]) == paths


@pytest.mark.parametrize("sr", [8000, 11025, 22050, 32000, 44100, 48000, 96000])
def test_native_resampler_matches_the_python_one_and_the_torchaudio_restatement(tmp_path, sr):
    """nomad.py:203-205: torchaudio.transforms.Resample(sr, 16000) defaults.  nomad_wav_read_rows against wavio.resample
    (numpy) and oracle/resample_oracle.py (torch conv1d over torchaudio's kernel): same length, samples within 1e-6."""
    import torch
    from oracle import resample_oracle as R
    rng = np.random.default_rng(sr)
    paths, raws = [], []
    for k, (ch, bits, n) in enumerate([(1, 16, sr + 123), (2, 24, sr // 3 + 7), (1, 32, 5), (1, 16, 40_001)]):
        x = np.clip(0.3 * rng.standard_normal((n, ch)), -0.99, 0.99)
        p = str(tmp_path / f"{k}.wav")
        open(p, "wb").write(_wav_bytes(x, sr, 1, bits))
        paths.append(p)
    info, status = wavio.probe(paths)
    assert status == [0] * len(paths)
    want = [wavio.load_processing(p)[0] for p in paths]                 # decode + mono + wavio.resample
    assert [wavio.frames_at(i) for i in info] == [w.shape[0] for w in want]
    host = np.zeros((len(paths), max(w.shape[0] for w in want) + 3), dtype=np.float32)
    wavio.read_rows(paths, list(info), range(len(paths)), host, threads=2)
    for r, (p, w) in enumerate(zip(paths, want)):
        got = host[r, :w.shape[0]]
        assert np.abs(got - w).max() < 1e-6, (sr, r)
        assert np.all(host[r, w.shape[0]:] == 0)
        raw, rate = wavio.read_wav(p)
        mono = ((raw[0] + raw[1]) / 2)[None] if raw.shape[0] > 1 else raw
        ref = R.resample(torch.from_numpy(mono), rate, 16000).numpy()[0]
        assert ref.shape == got.shape and np.abs(got - ref).max() < 1e-6, (sr, r)


def test_file_pipeline_native_raises_what_the_python_front_end_raises(tmp_path):
    x = np.zeros((500, 1))
    paths = []
    for i in range(30):
        p = str(tmp_path / f"{i:02d}.wav")
        open(p, "wb").write(_wav_bytes(x, 16000, 1, 16) if i != 17 else b"RIFFxxxxWAVEnope")
        paths.append(p)
    with pytest.raises(ValueError, match="17.wav"):
        _nomad(_Engine(), 2).get_embeddings_csv(None, pd.DataFrame({"filename": paths}), max_batch_samples=3000)
    with pytest.raises(FileNotFoundError):
        _nomad(_Engine(), 2).get_embeddings_csv(None, pd.DataFrame({"filename": paths[:5] + [str(tmp_path / "gone.wav")]}))


def test_staging_bound_closes_a_batch_when_one_long_file_sits_among_short_ones(tmp_path):
    lens = [300] * 20 + [20_000] + [300] * 20
    paths = []
    for i, n in enumerate(lens):
        p = str(tmp_path / f"{i:02d}.wav")
        open(p, "wb").write(_wav_bytes(np.zeros((n, 1)), 16000, 1, 16))
        paths.append(p)
    eng = _Engine()
    _nomad(eng, 2).get_embeddings_csv(None, pd.DataFrame({"filename": paths}), max_batch_samples=40_000)
    assert [l for b in eng.batches for l in b] == lens
    assert all(len(b) * max(b) <= 80_000 for b in eng.batches)           # rows x stride of every staging buffer


def test_reader_under_address_and_ub_sanitizers(tmp_path, wav_dir):
    """The host code of the reader (probe, decode, resample) built with g++ -fsanitize=address,undefined and run over the
    format zoo, truncated / malformed files and other sample rates, every output in an exactly-sized heap buffer; its
    sums must match the Python front end's."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "wav_reader_driver")
    subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-pthread",
                    os.path.join(root, "tests", "native", "wav_reader_driver.cpp"), "-o", exe], check=True, timeout=300)
    rng = np.random.default_rng(11)
    extra = []
    for k, (sr, ch, bits, n) in enumerate([(48000, 2, 16, 4801), (8000, 1, 24, 3), (44100, 1, 32, 12345), (22050, 3, 16, 1)]):
        p = str(tmp_path / f"r{k}.wav")
        open(p, "wb").write(_wav_bytes(np.clip(0.3 * rng.standard_normal((n, ch)), -0.99, 0.99), sr, 1, bits))
        extra.append(p)
    bad = []
    good = _wav_bytes(np.zeros((700, 2)), 16000, 1, 16)
    for k, data in enumerate([good[:50], good[:45], good[:13], b"RIFF\xff\xff\xff\xffWAVEfmt \xff\xff\xff\x7f", good[:len(good) - 3],
                              good.replace(b"data", b"dat_"), b""]):
        p = str(tmp_path / f"bad{k}.wav")
        open(p, "wb").write(data)
        bad.append(p)
    files = list(wav_dir) + extra + bad
    res = subprocess.run([exe] + files, capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert res.returncode == 0, res.stderr[-3000:]
    lines = res.stdout.strip().split("\n")
    assert len(lines) == len(files)
    for path, line in zip(files, lines):
        st, rate, ch, frames, n16, total = line.split()
        try:
            want = wavio.load_processing(path)[0]
        except (ValueError, struct.error):
            assert int(st) != 0, path
            continue
        assert int(st) == 0 and int(n16) == want.shape[0], (path, line)
        assert abs(float(total) - float(want.astype(np.float64).sum())) < 1e-3 + 1e-6 * want.shape[0], (path, line)
