"""Where a DeviceFeeder batch goes (GPU box): gather into pinned memory, H2D copy, alone and beside the train step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tacotron_amd.config import Config
from tacotron_amd.data import DeviceCorpus, DeviceFeeder, synthetic_corpus
from tacotron_amd.model import Tacotron
print('torch threads', torch.get_num_threads(), 'cpus', os.cpu_count())
corpus = synthetic_corpus(256, 200, 180, 2, 60)
idx = torch.as_tensor(np.random.default_rng(0).integers(256, size=32).astype(np.int64))
pinned = {k: torch.empty((32,) + tuple(v.shape[1:]), dtype=v.dtype, pin_memory=True) for k, v in corpus.items()}
dev = {k: torch.empty((32,) + tuple(v.shape[1:]), dtype=v.dtype, device='cuda') for k, v in corpus.items()}
for nt in (torch.get_num_threads(), 8, 1):
    torch.set_num_threads(nt)
    t0 = time.perf_counter()
    for _ in range(10):
        for k, v in corpus.items(): torch.index_select(v, 0, idx, out=pinned[k])
    print('gather into pinned, %d threads: %.2f ms' % (nt, (time.perf_counter() - t0) / 10 * 1e3))
torch.set_num_threads(8)
st = torch.cuda.Stream()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    with torch.cuda.stream(st):
        for k in corpus: dev[k].copy_(pinned[k], non_blocking=True)
    st.synchronize()
print('H2D of one batch from pinned: %.2f ms' % ((time.perf_counter() - t0) / 10 * 1e3))
for kind in (DeviceFeeder, DeviceCorpus):
    f = kind(corpus, 32, device='cuda')
    f.next(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): f.next()
    torch.cuda.synchronize()
    print('%s alone: %.2f ms per batch' % (kind.__name__, (time.perf_counter() - t0) / 50 * 1e3))
    c = Config(); c.r, c.vocab_size = 2, 60
    m = Tacotron(c, f.next(), train=True, seed=0)
    for _ in range(5): m.set_inputs(f.next()); m.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): m.set_inputs(f.next()); m.step()
    torch.cuda.synchronize()
    print('%s + train step: %.3f ms per step' % (kind.__name__, (time.perf_counter() - t0) / 100 * 1e3))
    f.close(); del m
