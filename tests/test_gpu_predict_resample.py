"""GPU: Nomad.predict end to end on files that are NOT 16 kHz mono - the reference's load_processing branch
(/root/reference/src/nomad_audio/nomad.py:196-205: two-channel mean, then torchaudio Resample(sr, 16000)) inside the product's
file pipeline (native reader -> pinned staging -> ragged batches -> HIP forward -> float64 distances), against the oracle:
oracle WAV reader -> the reference's down-mix -> oracle/resample_oracle.py (torchaudio's published algorithm) -> oracle forward
-> oracle pairwise.  Score tolerance 1e-4 (BASELINE.json north_star)."""
import os
import struct

import numpy as np
import pytest
import torch

from oracle import nomad_oracle as O
from oracle import resample_oracle as R

pytestmark = pytest.mark.gpu
SCORE_TOL = 1e-4


def _write_pcm16(path, x, sr):
    """x: (frames, channels) float in [-1, 1) -> RIFF/WAVE PCM-16."""
    pcm = np.clip(np.round(x * 32768), -32768, 32767).astype("<i2").tobytes()
    ch = x.shape[1]
    fmt = struct.pack("<HHIIHH", 1, ch, sr, sr * ch * 2, ch * 2, 16)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(pcm)) + pcm
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)


def _oracle_load(path):
    """nomad.py:196-205 on the oracle's reader: (channels, N) -> mean of channels 0 and 1 -> Resample(sr, 16000)."""
    x, sr = O.load_wav(path)
    w = torch.from_numpy(np.ascontiguousarray(x))
    if w.shape[0] > 1:
        w = ((w[0] + w[1]) / 2)[None, :]
    return R.resample(w, sr, 16000) if sr != 16000 else w


@pytest.mark.parametrize("native_threads", [4, 0], ids=["native_reader", "python_front_end"])
def test_predict_on_resampled_and_stereo_files_vs_oracle(built_lib, sd0, tmp_path, native_threads):
    from nomad_amd.nomad import Nomad
    rng = np.random.default_rng(11)
    nmr, deg = tmp_path / "nmr", tmp_path / "deg"
    nmr.mkdir(), deg.mkdir(), (tmp_path / "out").mkdir()   # like the reference, predict() writes into an EXISTING results_path
    t = lambda n, sr: np.arange(n) / sr                                                       # noqa: E731

    def tone(n, sr, ch, f0):     # band-limited content (below 3.5 kHz: representable at every rate used) + a little noise
        x = np.stack([0.3 * np.sin(2 * np.pi * (f0 + 70 * c) * t(n, sr)) + 0.2 * np.sin(2 * np.pi * 2.7 * f0 * t(n, sr) + c)
                      + 0.02 * rng.standard_normal(n) for c in range(ch)], axis=1)
        return np.clip(x, -0.99, 0.99)
    files = {  # name: (dir, sample rate, channels, seconds, f0)
        "ref48k_stereo": (nmr, 48000, 2, 1.3, 310.0), "ref8k_mono": (nmr, 8000, 1, 1.7, 220.0), "ref16k_mono": (nmr, 16000, 1, 1.1, 400.0),
        "deg48k_stereo": (deg, 48000, 2, 2.1, 330.0), "deg8k_mono": (deg, 8000, 1, 1.2, 250.0), "deg44k_stereo": (deg, 44100, 2, 1.5, 500.0),
        "deg16k_stereo": (deg, 16000, 2, 1.0, 180.0),
    }
    for name, (d, sr, ch, sec, f0) in files.items():
        _write_pcm16(str(d / f"{name}.wav"), tone(int(sec * sr), sr, ch, f0), sr)

    nmd = Nomad(weights=sd0)
    nmd.NATIVE_WAV_THREADS = native_threads
    df_avg, df_dm = nmd.predict("dir", str(nmr), str(deg), results_path=str(tmp_path / "out"))

    with torch.no_grad():
        emb = {name: O.triplet_forward(sd0, _oracle_load(str(d / f"{name}.wav")))[0].numpy() for name, (d, *_rest) in files.items()}
    ref_names = [n for n in files if files[n][0] is nmr]
    deg_names = [n for n in files if files[n][0] is deg]
    dref, mref = O.pairwise(np.stack([emb[n] for n in deg_names]), np.stack([emb[n] for n in ref_names]))
    # unrounded scores: the product's own embeddings through its float64 distance kernel
    e_deg = nmd.get_embeddings(str(deg)).set_index("filename")
    e_ref = nmd.get_embeddings(str(nmr)).set_index("filename")
    for n in deg_names:
        got = e_deg.loc[str(deg / f"{n}.wav")].to_numpy(dtype=np.float32)
        assert np.abs(got - emb[n]).max() < 2e-5, n            # embeddings: resampler (1e-6 on samples) + fp32 forward
    got_deg = torch.from_numpy(np.stack([e_deg.loc[str(deg / f"{n}.wav")].to_numpy(dtype=np.float32) for n in deg_names])).cuda()
    got_ref = torch.from_numpy(np.stack([e_ref.loc[str(nmr / f"{n}.wav")].to_numpy(dtype=np.float32) for n in ref_names])).cuda()
    d, m = nmd.engine.pairwise(got_deg, got_ref)
    assert np.abs(d.cpu().numpy() - dref).max() < SCORE_TOL and np.abs(m.cpu().numpy() - mref).max() < SCORE_TOL
    # the tables predict() returned (3 decimals, by name: listing order is the file system's)
    for i, dn in enumerate(deg_names):
        assert abs(df_avg.loc[dn, "NOMAD"] - round(float(mref[i]), 3)) <= 1.01e-3
        for j, rn in enumerate(ref_names):
            assert abs(df_dm.loc[dn, rn] - round(float(dref[i, j]), 3)) <= 1.01e-3
    assert float(np.abs(dref).min()) > 1e-3                    # the files really differ: no trivially-zero distances
    nmd.engine.close()
