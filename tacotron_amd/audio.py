"""r-frame layout of the reference's spectrogram tensors (audio.reshape_frames, audio.py:23-35) -- SURVEY §8f row 1.

The decoder emits r non-overlapping frames per step; the reference stores frames so that row `4c + j` of the
(steps, r*C) matrix holds frames `4rc + 4i + j` for i = 0..r-1 (C features each).  Written here as plain index
arithmetic (no split/concatenate chains); checked against vectors produced by the reference function itself
(tests/golden/reshape_frames.npz)."""
from __future__ import annotations

import numpy as np


def reshape_frames(signal, r, forward=True):
    signal = np.asarray(signal)
    if forward:
        C, T = signal.shape
        nch = T // (4 * r)                                  # only full chunks of 4r frames are kept
        x = signal[:, :nch * 4 * r].reshape(C, nch, r, 4)   # [ch, c, i, j] = signal[ch, 4rc + 4i + j]
        return x.transpose(1, 3, 2, 0).reshape(nch * 4, r * C)   # [4c + j, iC + ch]
    N, RC = signal.shape
    C = RC // r
    nch = N // 4
    x = signal[:nch * 4].reshape(nch, 4, r, C)              # [c, j, i, ch]
    return x.transpose(0, 2, 1, 3).reshape(nch * 4 * r, C)  # row 4rc + 4i + j


def denormalize(output, stft_mean, stft_std):
    """test.py:64 / train.py:94-95: out * stft_std + stft_mean."""
    return output * stft_std + stft_mean
