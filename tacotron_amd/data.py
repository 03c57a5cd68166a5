"""Host-side data helpers: synthetic Nancy-shaped batches and the reference's prompt front end
(data_input.py:87-108).  I/O of real corpora (data_input.load_from_npy, preprocess.py) is out of scope."""
from __future__ import annotations

import queue
import threading

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


def synthetic_corpus(n=256, Tt=200, Td=180, r=2, V=60, seed=1234, rank=0, num_speakers=1):
    """A pool of `n` synthetic utterances with the statistics of synthetic_batch, generated ONCE: the stand-in for
    the preprocessed npy corpus (data_input.py:43-85) when none is on disk.  Minibatches are index draws from the pool, as
    the reference's `slice_input_producer(shuffle=True)` draws from its arrays -- generating 51 MB of normal deviates per step
    on the host (synthetic_batch, ~150 ms) would be 18 x the 8.4 ms train step."""
    b = synthetic_batch(n, Tt, Td, r, V, seed=seed, rank=rank, num_speakers=num_speakers)
    b.pop('speech_length')
    return b


class DeviceCorpus:
    """The whole corpus resident in HBM (MI355X: 288 GB per GPU -- the Nancy corpus, 12 K utterances x 1.6 MB at the reference's
    padded shapes, is 19 GB), uploaded once; a minibatch is a device-side row gather on the caller's stream (51 MB: ~20 us of HBM
    time) and the train loop moves NO bytes over PCIe per step.  Same `next()` contract as DeviceFeeder, which remains the path
    for corpora beyond the budget."""

    def __init__(self, data, batch_size, device='cuda', seed=1000, draw=None, chunk_rows=64):
        self.device = torch.device(device)
        self.B = int(batch_size)
        host = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))) for k, v in data.items()}
        self.n = len(next(iter(host.values())))
        self._rng = np.random.default_rng(seed)
        self._draw = draw if draw is not None else (lambda step: self._rng.integers(self.n, size=self.B))
        self.data = {}
        for k, v in host.items():   # chunked upload: no second full-size pinned copy of a 19 GB array
            d = torch.empty(v.shape, dtype=v.dtype, device=self.device)
            for i in range(0, self.n, chunk_rows):
                d[i:i + chunk_rows].copy_(v[i:i + chunk_rows])
            self.data[k] = d
        self._step = 0

    @staticmethod
    def nbytes(data):
        return sum(int(np.prod(v.shape)) * (v.element_size() if isinstance(v, torch.Tensor) else v.itemsize) for v in data.values())

    def next(self):
        idx = torch.as_tensor(np.asarray(self._draw(self._step), dtype=np.int64)).to(self.device, non_blocking=True)
        self._step += 1
        return {k: torch.index_select(v, 0, idx) for k, v in self.data.items()}

    def close(self):
        pass


class DeviceFeeder:
    """Feeds a train loop whose step is shorter than one pageable host-to-device copy of its batch (51 MB at the Nancy shape:
    ~20 ms pageable, ~2 ms pinned): the replacement for the reference's queue runners (train.py:44-45, data_input.py:67-72).

    A worker thread draws the indices of batch s + depth, copies the rows into PINNED staging buffers and enqueues the copy into one of depth + 1 device buffer sets on a copy stream; `next()`
    makes the caller's stream wait for that copy's event (no host block) and hands out the device tensors -- `Tacotron.set_inputs`
    on them is a pointer swap.  A buffer set is overwritten only after the consumer's stream has passed the event recorded by the
    `next()` call that retired it.  With device='cpu' (tests) the same rotation runs synchronously without pinning."""

    def __init__(self, data, batch_size, device='cuda', depth=2, seed=1000, draw=None):
        self.data = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))) for k, v in data.items()}
        self.n = len(next(iter(self.data.values())))
        self.B, self.depth = int(batch_size), int(depth)
        self.device = torch.device(device)
        self.cuda = self.device.type == 'cuda'
        self._rng = np.random.default_rng(seed)
        self._draw = draw if draw is not None else (lambda step: self._rng.integers(self.n, size=self.B))
        nslots = self.depth + 1
        shp = {k: (self.B,) + tuple(v.shape[1:]) for k, v in self.data.items()}
        self._pinned = [{k: torch.empty(shp[k], dtype=v.dtype, pin_memory=self.cuda) for k, v in self.data.items()} for _ in range(nslots)]
        self._dev = [{k: torch.empty(shp[k], dtype=v.dtype, device=self.device) for k, v in self.data.items()} for _ in range(nslots)]
        self._ready = [None] * nslots          # copy-stream event: the slot's H2D copy is complete
        self._free = [None] * nslots           # consumer-stream event: the consumer no longer reads the slot
        self._copy = torch.cuda.Stream(self.device) if self.cuda else None
        self._q = queue.Queue()
        # a slot may be (re)filled once the batch that used it last has been retired by next() -- which is also when its
        # `_free` event exists; without this gate the worker would run one batch further and overwrite the set in use
        self._slots = threading.Semaphore(nslots)
        self._step = 0
        self._stop = False
        self._last = None
        self._dead = None                      # the exception the worker died with (re-raised by every later next())
        self._lock = threading.Lock()
        self._thread = threading.Thread(target=self._work, daemon=True)
        self._thread.start()

    def _fill(self, step):
        slot = step % (self.depth + 1)
        idx = [int(i) for i in np.asarray(self._draw(step)).reshape(-1)]
        for k, v in self.data.items():
            dst = self._pinned[slot][k]
            for i, j in enumerate(idx):      # row copies (a 1.4 MB memcpy each at the Nancy shape): no intra-op thread pool -- on a
                dst[i].copy_(v[j])           # 256-core host torch.index_select with its default 128 threads took 19 ms per batch, 0.5 ms with 8
        if self.cuda:
            with torch.cuda.stream(self._copy):
                if self._free[slot] is not None:
                    self._copy.wait_event(self._free[slot])
                for k in self.data:
                    self._dev[slot][k].copy_(self._pinned[slot][k], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._copy)
            self._ready[slot] = ev
            ev.synchronize()   # (worker thread only: the pinned slot may be refilled once its copy has left the host)
        else:
            for k in self.data:
                self._dev[slot][k].copy_(self._pinned[slot][k])
        return slot

    def _work(self):
        step = 0
        while True:
            self._slots.acquire()
            if self._stop:
                return
            try:
                slot = self._fill(step)
            except Exception as e:   # surfaces in next()
                self._q.put(e)
                return
            self._q.put(slot)
            step += 1

    def next(self):
        """Device tensors of the next batch.  Valid until the next call: kernels already enqueued on the caller's stream keep reading
        them safely (the refill waits for an event recorded on that stream), host-side readers must copy first."""
        if self._dead is not None:           # the worker is gone: every later call fails the same way instead of blocking forever
            raise self._dead
        while True:
            try:
                item = self._q.get(timeout=1.0)
                break
            except queue.Empty:
                if not self._thread.is_alive():   # died without reporting (never expected; never wait on a dead thread)
                    self._dead = RuntimeError('DeviceFeeder worker thread is not running')
                    raise self._dead
        if isinstance(item, Exception):
            self._dead = item
            raise item
        if self.cuda:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(self._ready[item])
        if self._last is not None:          # the slot handed out by the previous call is free once the consumer's stream gets here
            if self.cuda:
                ev = torch.cuda.Event()
                ev.record(cur)
                self._free[self._last] = ev
            self._slots.release()
        self._last = item
        self._step += 1
        return self._dev[item]

    def close(self):
        self._stop = True
        self._slots.release()
        self._thread.join(timeout=10.0)   # (the worker must not be inside a device call when the interpreter tears the runtime down)


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
