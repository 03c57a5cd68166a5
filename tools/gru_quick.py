"""bi-GRU launch times (HIP events, every launch by itself) + step time.  usage: [TACO_LIB=...] python tools/gru_quick.py"""
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
for _ in range(20): m.step()
e1.record(); torch.cuda.synchronize()
step = e0.elapsed_time(e1) / 20
lib.profile_read(3)
lib.profile_enable(0b11000)
N = 8
for _ in range(N): m.step()
torch.cuda.synchronize()
lib.profile_enable(0)
rm = lib.profile_read(3)
n = len(rm) // N
avg = [sum(rm[i + k * n] for k in range(N)) / N * 1e3 for i in range(n)]
print('%s step %.3f ms  bigru [enc fwd, post fwd, post bwd, enc bwd] = %s  sum %.1f us' % (
    os.path.basename(os.environ.get('TACO_LIB', 'libtaco_hip.so')), step, ['%.1f' % x for x in avg], sum(avg)), flush=True)
m.check()
