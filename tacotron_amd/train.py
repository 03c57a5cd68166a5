"""train.py -- host driver with the reference's CLI and loop contract (train.py:16-126): -t/--train-set, -d/--debug,
-r/--restore; LR annealing every 1000 steps, loss-explosion guard, checkpoint + listening sample every SAVE_EVERY steps
(train.py:85-103; the TensorBoard wrapping is out of scope, the sample is written as .wav / .npy under log/<save_path>/).
Without a preprocessed corpus under data/<set>/ it trains on synthetic Nancy-shaped batches."""
from __future__ import annotations

import argparse
import math
import os
import pickle as pkl

import numpy as np
import torch

from . import config as config_module
from .config import SAVE_EVERY, Config
from .data import DeviceCorpus, DeviceFeeder, synthetic_batch, synthetic_corpus
from .dist import GradReducer, init_from_env
from .model import Tacotron


def save_sample(model, out_dir, step, n_iter=50):
    """train.py:92-103: `ideal` = the target spectrogram of utterance 0 through Griffin-Lim, `sample` = the model's output for
    it, plus its alignment; both de-normalised with the corpus statistics (identity for synthetic data)."""
    from . import lib
    from .griffinlim import invert_spectrogram
    from .test import write_wav
    c = model.config
    os.makedirs(out_dir, exist_ok=True)
    dev = model.output.device
    mean = torch.as_tensor(model.stft_mean if model.stft_mean is not None else np.zeros(c.fft_size * c.r), dtype=torch.float32, device=dev)
    std = torch.as_tensor(model.stft_std if model.stft_std is not None else np.ones(c.fft_size * c.r), dtype=torch.float32, device=dev)
    pair = torch.stack([model.inputs['stft'][0], model.output[0]]).contiguous()       # (2, Td, 1025 r): ideal, sample
    wav = invert_spectrogram(pair, mean, std, c.r, n_iter=n_iter, seed=step).cpu().numpy()
    spec = lib.denorm_unframe(pair, mean, std, c.r).cpu().numpy()
    write_wav(os.path.join(out_dir, 'ideal_%d.wav' % step), wav[0])
    write_wav(os.path.join(out_dir, 'sample_%d.wav' % step), wav[1])
    np.save(os.path.join(out_dir, 'sample_%d_spec.npy' % step), spec[1])
    np.save(os.path.join(out_dir, 'attention_%d.npy' % step), model.alignments[0].cpu().numpy())


def load_corpus(data_path, seed=0):
    """data_input.load_meta + load_from_npy (data_input.py:43-85,110-113) when the npy files exist.  The normalisation
    statistics come from 100 utterances like the reference's (data_input.py:55-65) but from a SEEDED draw, so every rank
    (and a restarted run) standardises the targets identically; they are returned for the checkpoint (train.py:31-33)."""
    meta_path = os.path.join(data_path, 'meta.pkl')
    if not os.path.exists(meta_path):
        return None
    with open(meta_path, 'rb') as f:
        meta = pkl.load(f)
    arr = {n: np.load(os.path.join(data_path, n + '.npy')) for n in ('texts', 'text_lens', 'stfts', 'mels')}
    stft, mel = arr['stfts'].astype(np.float32), arr['mels'].astype(np.float32)
    idx = np.random.default_rng(seed).integers(len(stft), size=100)
    stft_mean, stft_std = stft[idx].mean((0, 1)), stft[idx].std((0, 1))
    mel_mean, mel_std = mel[idx].mean((0, 1)), mel[idx].std((0, 1))
    data = {'text': arr['texts'].astype(np.int32), 'text_length': arr['text_lens'].astype(np.int32),
            'stft': (stft - stft_mean) / stft_std, 'mel': (mel - mel_mean) / mel_std}
    spk_path = os.path.join(data_path, 'speakers.npy')   # data_input.py:76-83: present for multi-speaker corpora (VCTK)
    if os.path.exists(spk_path):
        data['speaker'] = np.load(spk_path).astype(np.int32)
    return meta, data, stft_mean, stft_std


def latest_checkpoint(ckpt_prefix):
    """tf.train.latest_checkpoint (train.py:49-52) for files named '<prefix>-<step>': the highest STEP, not the
    lexicographically last name ('tacotron-5000' sorts after 'tacotron-10000')."""
    d, base = os.path.dirname(ckpt_prefix) or '.', os.path.basename(ckpt_prefix)
    if not os.path.isdir(d):
        return None
    best, best_step = None, -1
    for f in os.listdir(d):
        head, sep, tail = f.rpartition('-')
        if sep and head == base and tail.isdigit() and int(tail) > best_step:
            best, best_step = os.path.join(d, f), int(tail)
    return best


def train(config, num_steps=1000000, log_every=50, save_every=SAVE_EVERY):
    rank, world, local = init_from_env()
    torch.cuda.set_device(local)
    corpus = load_corpus(config.data_path)
    stft_mean = stft_std = None
    if corpus is not None:
        meta, data, stft_mean, stft_std = corpus
        config.r, config.vocab_size = meta['r'], len(meta['vocab'])
        if 'speaker' in data:                       # train.py:29
            config.num_speakers = int(data['speaker'].max()) + 1
        n = len(data['text'])
    else:
        if rank == 0:
            print('no corpus under %s -- synthetic Nancy-shaped batches' % config.data_path)
        data, n = None, 0
    if data is None:   # a pool of synthetic utterances stands in for the npy corpus; batches are index draws from it either way
        data = synthetic_corpus(max(256, 4 * config.batch_size), 200, config.max_decode_iter, config.r, config.vocab_size,
                                seed=1234, rank=rank, num_speakers=config.num_speakers)
    # The reference's queue runners (train.py:44-45): the corpus lives in HBM and a batch is a device-side gather (DeviceCorpus);
    # a corpus beyond TACO_CORPUS_HBM_GB (default 64) is fed by a prefetching worker through pinned buffers and a copy stream
    # (DeviceFeeder).  Either way set_inputs() gets device tensors and is a pointer swap -- a blocking pageable copy of the 51 MB
    # batch alone would be ~2 x the 8.4 ms train step.
    dev = torch.device('cuda', local)
    budget = float(os.environ.get('TACO_CORPUS_HBM_GB', '64')) * (1 << 30)
    if DeviceCorpus.nbytes(data) <= budget:
        feeder = DeviceCorpus(data, config.batch_size, device=dev, seed=1000 + rank)
    else:
        feeder = DeviceFeeder(data, config.batch_size, device=dev, depth=2, seed=1000 + rank)

    def next_batch(step):
        return feeder.next()

    # same initial parameters on every rank (seed 0); the dropout / sampling streams are offset by the reducer's rank
    # TACO_FORCE_DIST=1 at world size 1 takes every distributed branch too (as bench.py does): reducer, collectives, barriers
    forced = torch.distributed.is_initialized() and world == 1
    distributed = world > 1 or forced
    model = Tacotron(config, next_batch(0), train=True, seed=0, reducer=GradReducer(force=forced) if distributed else None)
    model.stft_mean, model.stft_std = stft_mean, stft_std
    ckpt_prefix = os.path.join('weights', config.save_path)   # 'weights/nancy/tacotron' or 'weights/debug'
    if config.restore:
        # train.py:49-58: the latest checkpoint, unless the module constant RESTORE_FROM names a step -- then exactly
        # '<prefix>-<RESTORE_FROM>' (a missing file is an error there as well: saver.restore raises)
        restore_from = config_module.RESTORE_FROM
        path = latest_checkpoint(ckpt_prefix) if restore_from is None else '%s-%s' % (ckpt_prefix, restore_from)
        if path is not None:
            if restore_from is not None and not os.path.exists(path):
                raise FileNotFoundError('RESTORE_FROM=%s: %s does not exist' % (restore_from, path))
            model.load_state_dict(torch.load(path))
            if rank == 0:
                print('restored %s (global_step %d)' % (path, model.global_step))
    lr = config.init_lr
    import time
    t_mark, s_mark, step = None, 0, -1                    # throughput of the host loop itself (reported when the loop ends)
    try:
        for step in range(num_steps):
            model.set_inputs(next_batch(step))                # device tensors from the feeder: a pointer swap
            model.step(lr)
            gs = model.global_step
            if step == 10:                                    # (past the first steps' one-off costs: lazy allocations, LDS attribute calls)
                torch.cuda.synchronize()
                t_mark, s_mark = time.perf_counter(), step
            if gs % log_every == 0 or gs % save_every == 0:
                loss = float(model.loss)                      # the only host sync, every log_every steps
                try:
                    model.check()                             # decoder exchange time-outs surface here (sticky flag; the
                except Exception as e:                        # guarded Adam update skipped itself in the meantime)
                    if not getattr(e, 'recoverable', False):
                        raise
                    print('WARNING (rank %d): %s -- continuing' % (rank, e))   # check() moved to a more conservative decoder mode
                if rank == 0:
                    # tacotron.py:162-164: the summaries 'loss', 'seq2seq_loss', 'output_loss' (scalars only)
                    s2s_l, out_l = (float(x) for x in model.loss_terms)
                    print('step %d loss %.1f (seq2seq %.1f + output %.1f) gnorm %.2f decoder-mode %d' %
                          (gs, loss, s2s_l, out_l, float(model.global_gradient_norm), model.decoder_mode))
                # train.py:77-80; `not isfinite` added: the bf16x3 GEMMs turn an Inf operand into NaN (inf - inf in the plane split),
                # and `NaN > 1e8` is False -- a diverged run must still stop here
                if (not math.isfinite(loss) or loss > 1e8) and gs > 500:
                    print('loss exploded')
                    break
            if gs % 1000 == 0:
                lr *= config.annealing_rate                   # train.py:82-83
            if gs % save_every == 0 and gs != 0:
                if rank == 0:
                    print('saving weights')
                    os.makedirs(os.path.dirname(ckpt_prefix) or '.', exist_ok=True)
                    torch.save(model.state_dict(), '%s-%d' % (ckpt_prefix, gs))
                    print('saving sample')
                    save_sample(model, os.path.join('log', config.save_path), gs)
                if distributed:
                    # rank 0 spent a while on the host; the others must not run ahead into the next step's collectives (and the
                    # persistent decoder kernels of a rank that waits inside a collective keep spinning on their peers)
                    torch.distributed.barrier()
    except BaseException:
        #  a non-recoverable TacoError from check(), a feeder failure, Ctrl-C: the worker thread must not be inside a
        # device call when the interpreter tears the runtime down, and the process group must not outlive the loop
        try:
            torch.cuda.synchronize()
        except Exception:   # noqa: BLE001 -- the original exception is the one to report
            pass
        feeder.close()
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        raise
    torch.cuda.synchronize()
    feeder.close()
    if t_mark is not None and step > s_mark:
        dt = time.perf_counter() - t_mark
        fps = (step - s_mark) * config.batch_size * config.r * config.max_decode_iter * max(world, 1) / dt
        model.host_loop_frames_per_s = fps
        if rank == 0:
            print('host loop: %d steps in %.2f s = %.3f ms/step, %.0f mel-frames/s (%d rank%s), data fed by %s' %
                  (step - s_mark, dt, dt / (step - s_mark) * 1e3, fps, world, '' if world == 1 else 's', type(feeder).__name__))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
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
