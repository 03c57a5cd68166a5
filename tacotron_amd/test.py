"""test.py -- inference driver with the reference's contract (test.py:13-89): prompts on stdin -> batches of <= 32 padded
to 140 chars -> always max_decode_iter steps -> normalised log-magnitude spectrogram (B, Td, 1025 r) + alignments.
The Griffin-Lim vocoder / TensorBoard dump (test.py:60-69) is out of scope; results are saved as .npy."""
from __future__ import annotations

import argparse
import os
import pickle as pkl
import sys

import numpy as np
import torch

from .audio import reshape_frames
from .config import Config
from .data import load_prompts
from .model import Tacotron
from .params import ParamBuffer
from . import lib


def test(config, prompts, out_dir='log/test', checkpoint=None):
    meta_path = os.path.join(config.data_path, 'meta.pkl')
    if os.path.exists(meta_path):
        with open(meta_path, 'rb') as f:
            ivocab = pkl.load(f)['vocab']
    else:
        ivocab = {i + 1: ch for i, ch in enumerate("abcdefghijklmnopqrstuvwxyz '.,?!-")}
        ivocab[0] = '<pad>'
    config.vocab_size = len(ivocab)
    params = None
    os.makedirs(out_dir, exist_ok=True)
    n = 0
    for batch in load_prompts(prompts, ivocab):
        if params is None:
            shape = lib.make_shape(batch['text'].shape[0], batch['text'].shape[1], config.max_decode_iter, config.r,
                                   config.vocab_size)
            params = ParamBuffer(shape, 'cuda').init_(0)
            if checkpoint:
                params.flat.copy_(torch.load(checkpoint)['params'])
        model = Tacotron(config, batch, train=False, params=params)
        out, al = model.run()
        model.check()
        out, al = out.cpu().numpy(), al.cpu().numpy()
        for o, a_ in zip(out, al):
            np.save(os.path.join(out_dir, 'prompt_%03d_spec.npy' % n), reshape_frames(o, config.r, forward=False))
            np.save(os.path.join(out_dir, 'prompt_%03d_align.npy' % n), a_)
            n += 1
    print('wrote %d spectrograms to %s' % (n, out_dir))


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('-t', '--train-set', default='nancy')
    ap.add_argument('--checkpoint', default=None)
    a = ap.parse_args()
    prompts = [p for p in sys.stdin.readlines() if len(p) > 0]
    c = Config()
    c.data_path = 'data/%s/' % a.train_set
    c.save_path = a.train_set + '/tacotron'
    print('Building Tacotron')
    test(c, prompts, checkpoint=a.checkpoint)
