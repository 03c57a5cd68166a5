"""Generates tests/golden/reshape_frames.npz by importing the REFERENCE's audio.reshape_frames
(/root/reference/audio.py:23-35) in this container with tensorflow / librosa / tqdm stubbed in sys.modules
(the function itself is pure NumPy).  Only the resulting input/output vectors are committed -- the reference
source never travels.  Run from the repo root:  python tests/golden/make_reshape_golden.py
"""
import os
import sys
import types

import numpy as np

for name in ('tensorflow', 'librosa', 'tqdm'):
    m = types.ModuleType(name)
    if name == 'tqdm':
        m.tqdm = lambda x, **k: x
    sys.modules[name] = m
sys.path.insert(0, '/root/reference')
import audio  # noqa: E402  (reference module)

out = {}
rng = np.random.default_rng(7)
for r in (2, 5):
    audio.r = r
    for C, T in ((7, 8 * r * 5 + 1), (12, 361 if r == 2 else 4 * r * 9 + 3)):
        x = rng.standard_normal((C, T)).astype(np.float32)
        fwd = audio.reshape_frames(x)
        inv = audio.reshape_frames(fwd, forward=False)
        out['x_r%d_C%d' % (r, C)] = x
        out['fwd_r%d_C%d' % (r, C)] = fwd
        out['inv_r%d_C%d' % (r, C)] = inv
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'reshape_frames.npz')
np.savez_compressed(path, **out)
print(path, {k: v.shape for k, v in out.items()})
