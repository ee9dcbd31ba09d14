"""CPU oracle for the resampling step of ``Nomad.load_processing``.  TEST INFRASTRUCTURE ONLY.

/root/reference/src/nomad_audio/nomad.py:203-205 calls ``torchaudio.transforms.Resample(sr, 16000)`` with its
defaults.  torchaudio (pinned ``torchaudio==0.12.1`` in /root/reference/requirements.txt:3) is a third-party
dependency that is absent from /root/reference and from the build image, so its published algorithm is restated
here, written separately from the product's ``nomad_amd/wavio.py`` (numpy, einsum over strided windows): this file
follows torchaudio's own formulation - ``torchaudio.functional._get_sinc_resample_kernel`` (per-phase kernels of the
Hann-windowed sinc, ``resampling_method="sinc_interpolation"``, ``lowpass_filter_width=6``, ``rolloff=0.99``,
computed in float64 and cast to float32 as ``transforms.Resample`` does with ``dtype=None``) followed by
``_apply_sinc_resample_kernel`` (pad ``(width, width + orig)``, ``conv1d(stride=orig)``, interleave the phases, cut
to ``ceil(new * length / orig)``).  PARITY UNPINNED against torchaudio itself (not importable here); what the test
pins is that two independently written forms of the published algorithm agree to 1e-6.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """-> (kernels (new, 1, 2*width + orig) float32, width); frequencies already divided by their gcd."""
    base_freq = min(orig_freq, new_freq) * rolloff
    width = math.ceil(lowpass_filter_width * orig_freq / base_freq)
    idx = torch.arange(-width, width + orig_freq, dtype=torch.float64)
    kernels = []
    for i in range(new_freq):                       # one FIR per output phase
        t = (-i / new_freq + idx / orig_freq) * base_freq
        t = t.clamp(-lowpass_filter_width, lowpass_filter_width)
        window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
        t = t * math.pi
        kernel = torch.where(t == 0, torch.ones_like(t), torch.sin(t) / torch.where(t == 0, torch.ones_like(t), t))
        kernels.append(kernel * window)
    scale = base_freq / orig_freq
    return (torch.stack(kernels).view(new_freq, 1, -1) * scale).to(torch.float32), width


def resample(waveform: torch.Tensor, orig_freq: int, new_freq: int) -> torch.Tensor:
    """(channels, N) float32 -> (channels, ceil(N * new / orig)) like ``transforms.Resample(orig_freq, new_freq)``."""
    if orig_freq == new_freq:
        return waveform
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    kernel, width = sinc_resample_kernel(orig, new)
    num_wavs, length = waveform.shape
    x = F.pad(waveform, (width, width + orig))
    y = F.conv1d(x[:, None], kernel, stride=orig)           # (wavs, new, frames)
    y = y.transpose(1, 2).reshape(num_wavs, -1)               # phase-interleaved output samples
    return y[..., :int(math.ceil(new * length / orig))]
