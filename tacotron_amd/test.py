"""test.py -- inference driver with the reference's contract (test.py:13-89): prompts on stdin -> batches of <= 32 padded
to 140 chars -> always max_decode_iter steps -> normalised log-magnitude spectrogram (B, Td, 1025 r) + alignments ->
`audio.invert_spectrogram(out * stft_std + stft_mean)` per prompt (test.py:64).  The TensorBoard summary the reference
wraps each sample in (test.py:65-69) is out of scope; the sample itself is written as <out_dir>/prompt_NNN.wav (16 kHz, the
reference's sr) next to the de-normalised spectrogram and the alignment as .npy."""
from __future__ import annotations

import argparse
import os
import pickle as pkl
import sys
import wave

import numpy as np
import torch

from .config import Config
from .data import load_prompts
from .griffinlim import invert_spectrogram
from .model import Tacotron
from .params import ParamBuffer
from . import lib

SR = 16000   # test.py:11


def write_wav(path, samples, sr=SR):
    """librosa.output.write_wav's role (audio.py:73-74): mono PCM16, peak-normalised only if the signal would clip."""
    x = np.asarray(samples, dtype=np.float64)
    peak = np.max(np.abs(x)) if x.size else 0.0
    if peak > 1.0:
        x = x / peak
    with wave.open(path, 'wb') as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(sr)
        f.writeframes((x * 32767.0).astype('<i2').tobytes())


def test(config, prompts, out_dir='log/test', checkpoint=None, speaker=0, n_iter=50, vocode=True):
    """test.py:13-70: restore the checkpoint (weights AND stft_mean / stft_std, test.py:27-28), run every prompt batch,
    de-normalise `out * stft_std + stft_mean` (test.py:64), undo the r-frame layout and invert with Griffin-Lim -- all on the
    GPU (lib.denorm_unframe, tacotron_amd.griffinlim).  ONE Tacotron (workspace + outputs) serves every batch of the same
    size; only a smaller final batch builds a second one.  `speaker`: id fed to a multi-speaker model for every prompt
    (data_input.py:101-106 feeds none: the reference's test.py cannot drive its own VCTK model)."""
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
    models = {}   # batch size -> Tacotron
    os.makedirs(out_dir, exist_ok=True)
    n = 0
    for batch in load_prompts(prompts, ivocab):
        Bn = batch['text'].shape[0]
        if config.num_speakers > 1:
            batch['speaker'] = torch.full((Bn,), int(speaker), dtype=torch.int32)
        if params is None:
            shape = lib.make_shape(Bn, batch['text'].shape[1], config.max_decode_iter, config.r, config.vocab_size,
                                   config.num_speakers)
            params = ParamBuffer(shape, 'cuda').init_(0)
        model = models.get(Bn)
        if model is None:
            model = models[Bn] = Tacotron(config, batch, train=False, params=params)
            if ckpt is not None:
                model.load_state_dict(ckpt)
        else:
            model.set_inputs(batch)
        out, al = model.run()
        model.check()
        mean = model.stft_mean if model.stft_mean is not None else torch.zeros(config.fft_size * config.r)
        std = model.stft_std if model.stft_std is not None else torch.ones(config.fft_size * config.r)
        mean = torch.as_tensor(mean, dtype=torch.float32).cuda()
        std = torch.as_tensor(std, dtype=torch.float32).cuda()
        spec = lib.denorm_unframe(out, mean, std, config.r)                       # (B, Td*r, 1025) chronological log-magnitudes
        wav = invert_spectrogram(out, mean, std, config.r, n_iter=n_iter, seed=n).cpu().numpy() if vocode else None
        spec, al = spec.cpu().numpy(), al.cpu().numpy()
        for i in range(Bn):
            np.save(os.path.join(out_dir, 'prompt_%03d_spec.npy' % n), spec[i])
            np.save(os.path.join(out_dir, 'prompt_%03d_align.npy' % n), al[i])
            if wav is not None:
                write_wav(os.path.join(out_dir, 'prompt_%03d.wav' % n), wav[i])
            n += 1
    print('wrote %d samples to %s' % (n, out_dir))
    return n


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('-t', '--train-set', default='nancy')
    ap.add_argument('--checkpoint', default=None)
    ap.add_argument('--speaker', type=int, default=0, help='speaker id for a multi-speaker checkpoint')
    ap.add_argument('--out-dir', default='log/test')
    a = ap.parse_args()
    prompts = [p for p in sys.stdin.readlines() if len(p) > 0]
    c = Config()
    c.data_path = 'data/%s/' % a.train_set
    c.save_path = a.train_set + '/tacotron'
    print('Building Tacotron')
    test(c, prompts, out_dir=a.out_dir, checkpoint=a.checkpoint, speaker=a.speaker)
