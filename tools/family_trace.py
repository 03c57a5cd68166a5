"""Per-launch HIP-event times of the MFMA GEMM family over one train step (GPU box).  usage: python tools/family_trace.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_amd import lib
from tacotron_amd.config import Config
from tacotron_amd.data import synthetic_batch
from tacotron_amd.model import Tacotron
c = Config(); c.r, c.vocab_size = 2, 60
m = Tacotron(c, synthetic_batch(32, 200, 180, 2, 60), train=True, seed=0)
for _ in range(3): m.step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): m.step()
e1.record(); torch.cuda.synchronize()
print('step %.3f ms' % (e0.elapsed_time(e1) / 10))
lib.profile_read(2); lib.profile_read(3)
lib.profile_enable(0b11100)   # bit 4: side stream off, every launch timed by itself
N = 4
for _ in range(N): m.step()
torch.cuda.synchronize()
lib.profile_enable(0)
labels = lib.profile_labels(2)
ms, fl = lib.profile_read(2, with_flops=True)
n = len(ms) // N
tot = 0.0
print('%d launches per step' % n)
for i in range(n):
    t = sum(ms[i + k * n] for k in range(N)) / N
    tot += t
    print('#%2d %8.1f us %8.2f GF %6.1f TF  %s' % (i, t * 1e3, fl[i] / 1e9, fl[i] / (t * 1e-3) / 1e12 if t > 0 else 0, labels[i] if i < len(labels) else ''))
print('sum %.1f us' % (tot * 1e3))
rm = lib.profile_read(3)
print('bigru', ['%.1f' % (x * 1e3) for x in rm[:4]])
