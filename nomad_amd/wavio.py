"""WAV front end of ``Nomad.load_processing`` (/root/reference/src/nomad_audio/nomad.py:192-212)
without torchaudio: RIFF/WAVE decode -> fp32 in [-1, 1) -> mono -> 16 kHz.

* decode: PCM-16 / PCM-24 / PCM-32 scaled by 2^-(bits-1), IEEE float32/64 as is - the values
  ``torchaudio.load(normalize=True)`` returns (nomad.py:196).
* mono: ``(ch0 + ch1) / 2`` - the reference averages only the first two channels (nomad.py:199-200).
* resample: windowed-sinc polyphase resampler with torchaudio's ``transforms.Resample`` defaults
  (``sinc_interp_hann``, lowpass_filter_width=6, rolloff=0.99), only when sr != 16 kHz (nomad.py:203-205).
* trim to 10 s is off in predict (nomad.py:178) but kept as an option (nomad.py:208-210).
"""
from __future__ import annotations

import math
import os
import struct
from typing import Tuple

import numpy as np


def read_wav(path: str) -> Tuple[np.ndarray, int]:
    """-> (channels, frames) float32, sample rate."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt = 12, None
    while pos + 8 <= len(data):
        cid = data[pos:pos + 4]
        size = struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            tag, ch, sr, _, _, bits = struct.unpack("<HHIIHH", body[:16])
            if tag == 0xFFFE and len(body) >= 26:  # WAVE_FORMAT_EXTENSIBLE: real tag in the sub-format GUID
                tag = struct.unpack("<H", body[24:26])[0]
            fmt = (tag, ch, sr, bits)
        elif cid == b"data":
            if fmt is None:
                raise ValueError(f"{path}: data chunk before fmt chunk")
            tag, ch, sr, bits = fmt
            if tag == 1 and bits == 16:
                x = np.frombuffer(body, dtype="<i2").astype(np.float32) / 32768.0
            elif tag == 1 and bits == 24:
                raw = np.frombuffer(body[:len(body) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
                v = raw[:, 0] | (raw[:, 1] << 8) | (raw[:, 2] << 16)
                v = np.where(v & 0x800000, v - 0x1000000, v)
                x = v.astype(np.float32) / 8388608.0
            elif tag == 1 and bits == 32:
                x = (np.frombuffer(body, dtype="<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
            elif tag == 1 and bits == 8:
                x = (np.frombuffer(body, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
            elif tag == 3 and bits == 32:
                x = np.frombuffer(body, dtype="<f4").astype(np.float32)
            elif tag == 3 and bits == 64:
                x = np.frombuffer(body, dtype="<f8").astype(np.float32)
            else:
                raise ValueError(f"{path}: unsupported wav encoding (format tag {tag}, {bits} bits)")
            n = x.size // ch * ch
            return np.ascontiguousarray(x[:n].reshape(-1, ch).T), sr
        pos += 8 + size + (size & 1)
    raise ValueError(f"{path}: no data chunk")


def _sinc_kernel(orig: int, new: int, width: int = 6, rolloff: float = 0.99):
    """torchaudio ``_get_sinc_resample_kernel`` (sinc_interp_hann) in float64 -> (new, 1, K) float32."""
    g = math.gcd(orig, new)
    orig, new = orig // g, new // g
    base = min(orig, new) * rolloff
    w = int(math.ceil(width * orig / base))
    idx = np.arange(-w, w + orig, dtype=np.float64)[None, :] / orig
    t = np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx
    t = np.clip(t * base, -width, width)
    window = np.cos(t * math.pi / width / 2) ** 2
    t = t * math.pi
    scale = base / orig
    k = np.where(t == 0, 1.0, np.sin(t) / np.where(t == 0, 1.0, t)) * window * scale
    return k.astype(np.float32), w, orig, new


def resample(x: np.ndarray, sr: int, target: int = 16000) -> np.ndarray:
    """(channels, N) -> (channels, ceil(N*target/sr)), torchaudio.transforms.Resample defaults."""
    if sr == target:
        return x
    k, w, orig, new = _sinc_kernel(sr, target)
    n = x.shape[1]
    xp = np.pad(x.astype(np.float32), ((0, 0), (w, w + orig)))
    n_frames = (xp.shape[1] - k.shape[1]) // orig + 1
    win = np.lib.stride_tricks.sliding_window_view(xp, k.shape[1], axis=1)[:, ::orig][:, :n_frames]
    y = np.einsum("cfk,pk->cfp", win, k, optimize=True).reshape(x.shape[0], -1)
    target_len = int(math.ceil(new * n / orig))
    return np.ascontiguousarray(y[:, :target_len].astype(np.float32))


def load_processing(path, target_sr: int = 16000, trim: bool = False) -> np.ndarray:
    """``Nomad.load_processing``: file -> (1, N) float32 mono at 16 kHz."""
    if isinstance(path, np.ndarray):
        path = path[0]
    wave, sr = read_wav(str(path))
    if wave.shape[0] > 1:
        wave = ((wave[0, :] + wave[1, :]) / 2)[None, :]
    if sr != target_sr:
        wave = resample(wave, sr, target_sr)
        sr = target_sr
    if trim and wave.shape[1] > sr * 10:
        wave = wave[:, :sr * 10]
    return np.ascontiguousarray(wave, dtype=np.float32)


# ---- the C ABI's reader (include/nomad_hip.h: nomad_wav_probe / nomad_wav_read_rows) --------------------------------
def probe(paths, threads: int = 4):
    """Headers of many files on native threads -> (array of _lib.WavInfo, list of per-file status; 0 = decodable)."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    n = len(paths)
    arr = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
    info = (_lib.WavInfo * n)()
    status = (C.c_int * n)()
    _lib.check(lib.nomad_wav_probe(arr, n, info, status, int(threads)), "nomad_wav_probe")
    return info, list(status)


def frames_at(info, target_sr: int = 16000) -> int:
    """Samples a probed file has at ``target_sr`` (after resampling, if its rate differs)."""
    import ctypes as C
    from . import _lib
    n = C.c_longlong()
    _lib.check(_lib.load().nomad_wav_frames_at(C.byref(info), int(target_sr), C.byref(n)), "nomad_wav_frames_at")
    return int(n.value)


def read_rows(paths, infos, rows, host, threads: int = 4, target_sr: int = 16000) -> None:
    """Sample data of probed files -> ``host[rows[i], :frames_i]`` (a 2-D contiguous fp32 torch tensor or numpy array),
    mono fp32 at ``target_sr`` as ``load_processing`` returns it (bit-identical without resampling, to fp32 summation
    order with)."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    n = len(paths)
    if n == 0:
        return
    if isinstance(host, np.ndarray):
        assert host.dtype == np.float32 and host.ndim == 2 and host.flags.c_contiguous
        ptr, stride = host.ctypes.data, host.shape[1]
    else:
        assert host.dtype.is_floating_point and host.element_size() == 4 and host.dim() == 2 and host.is_contiguous()
        ptr, stride = host.data_ptr(), host.shape[1]
    arr = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
    inf = (_lib.WavInfo * n)(*infos)
    row = (C.c_int * n)(*[int(r) for r in rows])
    status = (C.c_int * n)()
    _lib.check(lib.nomad_wav_read_rows(arr, inf, n, row, ptr, stride, int(target_sr), status, int(threads)), "nomad_wav_read_rows")
