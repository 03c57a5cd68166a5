"""Vocoder boundary of the reference (audio.invert_spectrogram / audio.griffinlim, audio.py:69-97; test.py:64) on the GPU.

`invert_spectrogram(out, stft_mean, stft_std, r)` takes what `Tacotron.run()` returns -- (B, Td, 1025 r) normalised
log-magnitude frames in the r-frame layout -- and returns waveforms (B, 300 (F - 1)), F = (Td // 4) * 4 * r:
de-normalise + inverse r-frame layout + exp + transpose in ONE HIP gather (taco_denorm_unframe), then Griffin-Lim
(taco_griffinlim: hand-written 2048-point FFT, 50 rounds like the reference).  The reference draws the initial phase with
np.random.rand; here it comes from a seeded torch generator so that a run can be reproduced."""
from __future__ import annotations

import math

import torch

from . import lib


def invert_spectrogram(out, stft_mean, stft_std, r, n_iter=50, seed=0, phase0=None):
    dev = out.device
    mean = torch.as_tensor(stft_mean, dtype=torch.float32, device=dev)
    std = torch.as_tensor(stft_std, dtype=torch.float32, device=dev)
    mag_t = lib.denorm_unframe(out.contiguous(), mean, std, r, want_spec=False, want_mag_t=True)   # (B, 1025, F)
    if phase0 is None:
        g = torch.Generator(device='cpu').manual_seed(seed)
        phase0 = (2.0 * math.pi * torch.rand(mag_t.shape, generator=g)).to(dev)
    return lib.griffinlim(mag_t, phase0.contiguous(), n_iter)
