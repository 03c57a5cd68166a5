"""Host-side data helpers: synthetic Nancy-shaped batches and the reference's prompt front end
(data_input.py:87-108).  I/O of real corpora (data_input.load_from_npy, preprocess.py) is out of scope."""
from __future__ import annotations

import numpy as np
import torch

from .config import MAX_TEXT_LEN


def synthetic_batch(B=32, Tt=200, Td=180, r=2, V=60, seed=1234, rank=0, min_len=50, num_speakers=1):
    """text_length ~ U{min_len..Tt} with row 0 forced to Tt; ids ~ U{1..V-1} inside the length, 0 (pad)
    outside; mel/stft ~ N(0,1) (the reference standardises per feature, data_input.py:55-65); speech_length = Td."""
    rng = np.random.default_rng(seed + rank)
    lo = min(min_len, Tt)
    tl = rng.integers(lo, Tt + 1, size=B).astype(np.int32)
    tl[0] = Tt
    text = rng.integers(1, V, size=(B, Tt)).astype(np.int32)
    text[np.arange(Tt)[None, :] >= tl[:, None]] = 0
    mel = rng.standard_normal((B, Td, 80 * r), dtype=np.float32)
    stft = rng.standard_normal((B, Td, 1025 * r), dtype=np.float32)
    out = {
        'text': torch.from_numpy(text), 'text_length': torch.from_numpy(tl),
        'mel': torch.from_numpy(mel), 'stft': torch.from_numpy(stft),
        'speech_length': torch.full((B,), Td, dtype=torch.int32),
    }
    if num_speakers > 1:   # VCTK-shaped: speaker ~ U{0..S-1}
        out['speaker'] = torch.from_numpy(rng.integers(0, num_speakers, size=B).astype(np.int32))
    return out


def pad(text, max_len, pad_val):
    """data_input.pad (data_input.py:87-90)."""
    return np.array([np.pad(np.asarray(t, dtype=np.int64), (0, max_len - len(t)), 'constant', constant_values=pad_val)
                     for t in text], dtype=np.int32)


class Vocab:
    """preprocess.py:19-22,130-135: the character vocabulary grows in first-seen order, '<pad>' is id 0.
    `process_char` is the reference's function of the same name; `ivocab` is what `load_prompts` consumes."""

    def __init__(self):
        self.vocab = {'<pad>': 0}
        self.ivocab = {0: '<pad>'}

    def process_char(self, char):
        if char not in self.vocab:
            nxt = len(self.vocab)
            self.vocab[char] = nxt
            self.ivocab[nxt] = char
        return self.vocab[char]

    def encode(self, prompt):
        return [self.process_char(ch) for ch in prompt]


def pad_to_dense(inputs):
    """preprocess.pad_to_dense (preprocess.py:137-146): ragged 1-D / 2-D arrays -> one zero-padded dense stack."""
    max_len = max(r.shape[0] for r in inputs)
    if inputs[0].ndim == 1:
        padded = [np.pad(a, (0, max_len - a.shape[0]), 'constant', constant_values=0) for a in inputs]
    else:
        padded = [np.pad(a, ((0, max_len - a.shape[0]), (0, 0)), 'constant', constant_values=0) for a in inputs]
    return np.stack(padded)


def load_prompts(prompts, ivocab, batch_size=32):
    """data_input.load_prompts (data_input.py:92-108) without the TF queue: unknown characters are dropped,
    `text_length = len(raw line)` (it counts the newline and dropped characters -- reproduced as is), prompts are padded
    to MAX_TEXT_LEN=140, and batches of <= 32 are yielded with a smaller final batch."""
    vocab = {v: k for k, v in ivocab.items()}
    text = [[vocab[w] for w in p.strip() if w in vocab] for p in prompts]
    text_length = np.array([len(p) for p in prompts], dtype=np.int32)
    text = pad(text, MAX_TEXT_LEN, 0)
    for i in range(0, len(prompts), batch_size):
        yield {'text': torch.from_numpy(text[i:i + batch_size].copy()),
               'text_length': torch.from_numpy(np.minimum(text_length[i:i + batch_size], MAX_TEXT_LEN))}
