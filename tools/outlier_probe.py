"""Root-cause probe for the sporadic slow inference call.  Runs N inference calls, each bracketed by CUDA events
and host timers, with the decoder launch timed by the library's own event ring, and prints every call whose wall time exceeds
2x the median together with (a) GPU time between the bracketing events, (b) the decoder kernel's own duration in that call,
(c) host time spent ENQUEUEING the call (before any synchronisation).  A persistent-kernel problem shows in (b); a GPU-side
stall outside the decoder in (a) - (b); a host / driver stall in (c) with (a) large only because the stream starved.
usage: python tools/outlier_probe.py [B] [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tacotron_amd import lib
from tacotron_amd.config import Config
from tacotron_amd.data import synthetic_batch
from tacotron_amd.model import Tacotron
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 400
ci = Config(); ci.r, ci.vocab_size, ci.max_decode_iter = 2, 60, 180
mi = Tacotron(ci, synthetic_batch(B, 140, 180, 2, 60, seed=77, min_len=40), train=False, seed=0)
for _ in range(3): mi.run()
torch.cuda.synchronize()
lib.profile_read(0); lib.profile_enable(1)
rows = []
for i in range(N):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record(); mi.run(); e1.record(); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    rows.append(((t2 - t0) * 1e3, e0.elapsed_time(e1), (t1 - t0) * 1e3))
    if i % 100 == 99:
        dec = lib.profile_read(0)
        for j, d in enumerate(dec): rows[i - len(dec) + 1 + j] = rows[i - len(dec) + 1 + j] + (d,)
lib.profile_enable(0)
mi.check()
a = np.array([r for r in rows if len(r) == 4])
med = np.median(a, 0)
print('B=%d, %d calls: median wall %.2f ms, GPU span %.2f ms, host enqueue %.2f ms, decoder kernel %.3f ms; decoder kernel max %.3f ms' %
      (B, len(a), med[0], med[1], med[2], med[3], a[:, 3].max()))
out = np.nonzero(a[:, 0] > 2 * med[0])[0]
print('%d calls slower than 2x the median:' % len(out))
for i in out[:20]:
    print('  call %4d: wall %.2f ms, GPU span %.2f ms, host enqueue %.2f ms, decoder kernel %.3f ms' % (i, *a[i]))

# ---- the sequence in which the stall was seen: a FRESH model object, two warm-up calls, then timed calls ----
print('fresh-model sequences (wall / GPU span / host enqueue, ms) of the first 5 timed calls after 2 warm-ups:')
worst = 0.0
for rep in range(12):
    Bi = (1, 32)[rep % 2]
    m2 = Tacotron(ci, synthetic_batch(Bi, 140, 180, 2, 60, seed=77 + rep, min_len=40), train=False, seed=0)
    for _ in range(2): m2.run()
    torch.cuda.synchronize()
    seq = []
    for i in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record(); m2.run(); e1.record(); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        seq.append(((t2 - t0) * 1e3, e0.elapsed_time(e1), (t1 - t0) * 1e3))
    worst = max(worst, max(s[0] for s in seq))
    print('  rep %2d B=%2d: %s' % (rep, Bi, '  '.join('%.1f/%.1f/%.2f' % s for s in seq)))
    del m2
print('worst wall of a timed call: %.1f ms' % worst)
