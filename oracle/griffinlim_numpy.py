"""CPU restatement of the reference's vocoder path -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference/audio.py:69-97 (`invert_spectrogram`, `griffinlim`: 50 rounds of librosa.istft / librosa.stft with
n_fft = 2048, win_length = 1200, hop_length = 300, window 'hann', random initial phase).

PARITY UNPINNED: the arithmetic lives in the un-vendored, un-pinned dependency `librosa` (README.md:20; absent from this
image, no network).  `stft` / `istft` below restate librosa's published algorithm of that era (0.5 / 0.6):
  stft : y reflect-padded by n_fft/2 (center=True), frames of n_fft at hop 300, periodic Hann(win_length) zero-padded to
         n_fft around its centre (util.pad_center), numpy rfft                            (librosa/core/spectrum.py stft)
  istft: irfft of every column, times the same padded window, overlap-added at hop 300 into n_fft + hop (n_frames - 1)
         samples, divided by filters.window_sumsquare where that exceeds `tiny`, then the n_fft/2 padding trimmed (istft)
What pins it here: istft(stft(y)) == y to fp64 round-off on the interior (tests/test_oracle.py), an independent
cross-check of `stft` against scipy.signal.stft with matching conventions, and the hand-written DFT of a short frame.
`griffinlim` takes the initial angles as an argument (the reference draws them with np.random.rand, audio.py:81)."""
from __future__ import annotations

import numpy as np

N_FFT = 2048
WIN = 1200
HOP = 300


def hann_periodic(m=WIN):
    """scipy.signal.get_window('hann', m, fftbins=True)."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(m) / m)


def pad_center(w, n=N_FFT):
    lpad = (n - len(w)) // 2
    return np.pad(w, (lpad, n - len(w) - lpad))


def window_sumsquare(n_frames, n_fft=N_FFT, win=WIN, hop=HOP):
    n = n_fft + hop * (n_frames - 1)
    x = np.zeros(n)
    wsq = pad_center(hann_periodic(win) ** 2, n_fft)
    for i in range(n_frames):
        s = i * hop
        x[s:min(n, s + n_fft)] += wsq[:max(0, min(n_fft, n - s))]
    return x


def stft(y, n_fft=N_FFT, win=WIN, hop=HOP):
    w = pad_center(hann_periodic(win), n_fft)
    yp = np.pad(np.asarray(y, dtype=np.float64), n_fft // 2, mode='reflect')
    n_frames = 1 + (len(yp) - n_fft) // hop
    out = np.empty((1 + n_fft // 2, n_frames), dtype=np.complex128)
    for t in range(n_frames):
        out[:, t] = np.fft.rfft(w * yp[t * hop:t * hop + n_fft])
    return out


def istft(S, n_fft=N_FFT, win=WIN, hop=HOP):
    n_frames = S.shape[1]
    w = pad_center(hann_periodic(win), n_fft)
    y = np.zeros(n_fft + hop * (n_frames - 1))
    for i in range(n_frames):
        y[i * hop:i * hop + n_fft] += w * np.fft.irfft(S[:, i], n_fft)
    wss = window_sumsquare(n_frames, n_fft, win, hop)
    nz = wss > np.finfo(np.float64).tiny
    y[nz] /= wss[nz]
    return y[n_fft // 2:-(n_fft // 2)]


def griffinlim(mag, angles0, n_iter=50):
    """audio.griffinlim (audio.py:77-97).  mag (1025, F) >= 0; angles0 (1025, F) radians (the reference: 2 pi rand)."""
    mag = np.abs(np.asarray(mag, dtype=np.float64))
    angles = np.exp(1j * np.asarray(angles0, dtype=np.float64))
    for _ in range(n_iter):
        inverse = istft(mag * angles)
        rebuilt = stft(inverse)
        angles = np.exp(1j * np.angle(rebuilt))
    return istft(mag * angles)


def spectral_convergence(y, mag):
    """|| |STFT(y)| - mag ||_F / || mag ||_F -- the quantity the reference prints when verbose (audio.py:90-92)."""
    return float(np.linalg.norm(np.abs(stft(y)) - mag) / np.linalg.norm(mag))

