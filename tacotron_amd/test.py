"""test.py -- inference driver with the reference's contract (test.py:13-89): prompts on stdin -> batches of <= 32 padded
to 140 chars -> always max_decode_iter steps -> normalised log-magnitude spectrogram (B, Td, 1025 r) + alignments.
The TensorBoard dump (test.py:60-69) is out of scope; de-normalised spectrograms are saved as .npy
(the Griffin-Lim inversion of audio.py:77-97 is tacotron_amd.griffinlim)."""
from __future__ import annotations

import argparse
import os
import pickle as pkl
import sys

import numpy as np
import torch

from .config import Config
from .data import load_prompts
from .model import Tacotron
from .params import ParamBuffer
from . import lib


def test(config, prompts, out_dir='log/test', checkpoint=None):
    """test.py:13-70: restore the checkpoint (weights AND stft_mean / stft_std, test.py:27-28), run every prompt batch,
    de-normalise `out * stft_std + stft_mean` (test.py:64) and undo the r-frame layout -- both on the GPU
    (lib.denorm_unframe) -- then hand the log-magnitude spectrogram to the vocoder (tacotron_amd.griffinlim, also HIP)."""
    meta_path = os.path.join(config.data_path, 'meta.pkl')
    if os.path.exists(meta_path):
        with open(meta_path, 'rb') as f:
            ivocab = pkl.load(f)['vocab']
    else:
        ivocab = {i + 1: ch for i, ch in enumerate("abcdefghijklmnopqrstuvwxyz '.,?!-")}
        ivocab[0] = '<pad>'
    config.vocab_size = len(ivocab)
    ckpt = torch.load(checkpoint) if checkpoint else None
    if ckpt is not None:
        config.r, config.vocab_size = ckpt.get('shape', (config.r, config.vocab_size))
        config.num_speakers = int(ckpt.get('num_speakers', 1))
    params = None
    os.makedirs(out_dir, exist_ok=True)
    n = 0
    for batch in load_prompts(prompts, ivocab):
        if config.num_speakers > 1:   # data_input.py:101-106 feeds no speaker for prompts; speaker 0 unless the caller chose one
            batch['speaker'] = torch.full((batch['text'].shape[0],), int(getattr(config, 'test_speaker', 0)), dtype=torch.int32)
        if params is None:
            shape = lib.make_shape(batch['text'].shape[0], batch['text'].shape[1], config.max_decode_iter, config.r,
                                   config.vocab_size, config.num_speakers)
            params = ParamBuffer(shape, 'cuda').init_(0)
        model = Tacotron(config, batch, train=False, params=params)
        if ckpt is not None:
            model.load_state_dict(ckpt)
        out, al = model.run()
        model.check()
        mean = model.stft_mean if model.stft_mean is not None else torch.zeros(config.fft_size * config.r)
        std = model.stft_std if model.stft_std is not None else torch.ones(config.fft_size * config.r)
        spec = lib.denorm_unframe(out, torch.as_tensor(mean, dtype=torch.float32).cuda(),
                                  torch.as_tensor(std, dtype=torch.float32).cuda(), config.r)   # (B, Td*r, 1025)
        spec, al = spec.cpu().numpy(), al.cpu().numpy()
        for o, a_ in zip(spec, al):
            np.save(os.path.join(out_dir, 'prompt_%03d_spec.npy' % n), o)
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
