"""train.py -- host driver with the reference's CLI and loop contract (train.py:16-126): -t/--train-set, -d/--debug,
-r/--restore; LR annealing every 1000 steps, loss-explosion guard, checkpoint every SAVE_EVERY steps.
Without a preprocessed corpus under data/<set>/ it trains on synthetic Nancy-shaped batches."""
from __future__ import annotations

import argparse
import os
import pickle as pkl

import numpy as np
import torch

from .config import SAVE_EVERY, Config
from .data import synthetic_batch
from .dist import GradReducer, init_from_env
from .model import Tacotron


def load_corpus(data_path):
    """data_input.load_meta + load_from_npy (data_input.py:43-85,110-113) when the npy files exist."""
    meta_path = os.path.join(data_path, 'meta.pkl')
    if not os.path.exists(meta_path):
        return None
    with open(meta_path, 'rb') as f:
        meta = pkl.load(f)
    arr = {n: np.load(os.path.join(data_path, n + '.npy')) for n in ('texts', 'text_lens', 'stfts', 'mels')}
    stft, mel = arr['stfts'].astype(np.float32), arr['mels'].astype(np.float32)
    idx = np.random.randint(len(stft), size=100)
    stft_mean, stft_std = stft[idx].mean((0, 1)), stft[idx].std((0, 1))
    mel_mean, mel_std = mel[idx].mean((0, 1)), mel[idx].std((0, 1))
    return meta, {'text': arr['texts'].astype(np.int32), 'text_length': arr['text_lens'].astype(np.int32),
                  'stft': (stft - stft_mean) / stft_std, 'mel': (mel - mel_mean) / mel_std}, stft_mean, stft_std


def train(config, num_steps=1000000, log_every=50):
    rank, world, local = init_from_env()
    torch.cuda.set_device(local)
    corpus = load_corpus(config.data_path)
    if corpus is not None:
        meta, data, stft_mean, stft_std = corpus
        config.r, config.vocab_size = meta['r'], len(meta['vocab'])
        n = len(data['text'])
    else:
        if rank == 0:
            print('no corpus under %s -- synthetic Nancy-shaped batches' % config.data_path)
        data, n = None, 0

    def next_batch(step):
        if data is None:
            return synthetic_batch(config.batch_size, 200, config.max_decode_iter, config.r, config.vocab_size,
                                   seed=1234 + step * 9973, rank=rank)
        idx = np.random.randint(n, size=config.batch_size)
        return {k: torch.from_numpy(v[idx]) for k, v in data.items()}

    model = Tacotron(config, next_batch(0), train=True, seed=0, reducer=GradReducer() if world > 1 else None)
    ckpt_dir = os.path.join('weights', config.save_path)
    if config.restore:
        cands = sorted(f for f in os.listdir(os.path.dirname(ckpt_dir) or '.') if f.startswith('tacotron-')) \
            if os.path.isdir(os.path.dirname(ckpt_dir)) else []
        if cands:
            model.load_state_dict(torch.load(os.path.join(os.path.dirname(ckpt_dir), cands[-1])))
    lr = config.init_lr
    for step in range(num_steps):
        model.set_inputs(next_batch(step))
        model.step(lr)
        gs = model.global_step
        if gs % log_every == 0 or gs % SAVE_EVERY == 0:
            loss = float(model.loss)                      # the only host sync, every log_every steps
            model.check()                                 # decoder exchange time-outs surface here
            if rank == 0:
                print('step %d loss %.1f gnorm %.2f' % (gs, loss, float(model.global_gradient_norm)))
            if loss > 1e8 and gs > 500:                   # train.py:77-80
                print('loss exploded')
                break
        if gs % 1000 == 0:
            lr *= config.annealing_rate                   # train.py:82-83
        if gs % SAVE_EVERY == 0 and gs != 0 and rank == 0:
            os.makedirs(os.path.dirname(ckpt_dir) or '.', exist_ok=True)
            torch.save(model.state_dict(), '%s-%d' % (ckpt_dir, gs))
    return model


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('-t', '--train-set', default='nancy')
    ap.add_argument('-d', '--debug', type=bool, default=False)
    ap.add_argument('-r', '--restore', type=bool, default=False)
    ap.add_argument('--steps', type=int, default=1000000)
    a = ap.parse_args()
    c = Config()
    c.data_path = 'data/%s/' % a.train_set
    c.restore = a.restore
    c.save_path = 'debug' if a.debug else '%s/tacotron' % a.train_set
    print('Building Tacotron')
    train(c, a.steps)
