"""Generates tests/golden/text_frontend.npz by importing the REFERENCE's pure-NumPy host helpers in this container with
tensorflow / librosa / tqdm / matplotlib stubbed in sys.modules:
  data_input.pad (data_input.py:87-90), preprocess.process_char / pad_to_dense (preprocess.py:130-146).
Only the resulting input/output vectors are committed -- the reference source never travels.
Run from the repo root:  python tests/golden/make_text_golden.py
"""
import os
import sys
import types

import numpy as np

for name in ('tensorflow', 'librosa', 'tqdm', 'matplotlib', 'matplotlib.pyplot'):
    m = types.ModuleType(name)
    if name == 'tqdm':
        m.tqdm = lambda x, **k: x
    if name == 'matplotlib':
        m.use = lambda *a, **k: None
    sys.modules[name] = m
sys.modules['matplotlib'].pyplot = sys.modules['matplotlib.pyplot']
sys.path.insert(0, '/root/reference')
import data_input  # noqa: E402  (reference module)
import preprocess  # noqa: E402  (reference module)

rng = np.random.default_rng(11)
out = {}
# data_input.pad: ragged int lists -> (n, max_len) int32 with a pad value
lens = [0, 3, 7, 1, 12]
rows = [rng.integers(1, 50, size=n).tolist() for n in lens]
out['pad_flat'] = np.array([v for r_ in rows for v in r_], dtype=np.int64)
out['pad_lens'] = np.array(lens, dtype=np.int64)
out['pad_out_140_0'] = data_input.pad(rows, 140, 0)
out['pad_out_12_9'] = data_input.pad(rows, 12, 9)
# preprocess.process_char: incremental vocabulary ('<pad>' = 0 pre-registered), ids in first-seen order
prompts = ['Hello, world.', 'the quick brown fox', 'HELLO again; 123!']
ids = [[preprocess.process_char(ch) for ch in p] for p in prompts]
out['char_ids_flat'] = np.array([v for r_ in ids for v in r_], dtype=np.int64)
out['char_lens'] = np.array([len(r_) for r_ in ids], dtype=np.int64)
out['prompts'] = np.array(prompts)
items = sorted(preprocess.vocab.items(), key=lambda kv: kv[1])
out['vocab_chars'] = np.array([k for k, _ in items])
out['vocab_ids'] = np.array([v for _, v in items], dtype=np.int64)
# preprocess.pad_to_dense: ragged 1-D (texts) and 2-D (frames x features) -> zero-padded dense stacks
texts = [np.array(r_, dtype=np.int32) for r_ in ids]
out['dense_1d'] = preprocess.pad_to_dense(texts)
mats = [rng.standard_normal((n, 5)).astype(np.float32) for n in (4, 9, 2)]
for i, m_ in enumerate(mats):
    out['mat_%d' % i] = m_
out['dense_2d'] = preprocess.pad_to_dense(mats)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'text_frontend.npz')
np.savez_compressed(path, **out)
print(path, {k: getattr(v, 'shape', None) for k, v in out.items()})
