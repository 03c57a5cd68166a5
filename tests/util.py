import numpy as np


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))


def max_abs(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))


def report(name, a, b):
    r, m = rel_l2(a, b), max_abs(a, b)
    print('  %-40s rel_l2=%.3e max_abs=%.3e  |ref|inf=%.3e' % (name, r, m, float(np.max(np.abs(b)))))
    return r, m


def small_case(r=2, V=20, B=2, Tt=9, Td=5, seed=3, full_len_row0=True):
    """Seeded tiny inputs + masks for oracle <-> HIP comparisons (numpy, fp64 targets rounded to fp32)."""
    rng = np.random.default_rng(seed)
    text = rng.integers(1, V, size=(B, Tt)).astype(np.int32)
    tl = rng.integers(max(1, Tt // 2), Tt + 1, size=B).astype(np.int32)
    if full_len_row0:
        tl[0] = Tt
    for b in range(B):
        text[b, tl[b]:] = 0
    mel = rng.standard_normal((B, Td, 80 * r)).astype(np.float32)
    stft = rng.standard_normal((B, Td, 1025 * r)).astype(np.float32)
    masks = {
        'enc_keep1': rng.integers(0, 2, (B, Tt, 256)).astype(np.uint8),
        'enc_keep2': rng.integers(0, 2, (B, Tt, 128)).astype(np.uint8),
        'dec_keep1': rng.integers(0, 2, (B, Td, 256)).astype(np.uint8),
        'dec_keep2': rng.integers(0, 2, (B, Td, 128)).astype(np.uint8),
        'sample': rng.integers(0, 2, (Td, B)).astype(np.uint8),
    }
    return {'text': text, 'text_length': tl, 'mel': mel, 'stft': stft}, masks
