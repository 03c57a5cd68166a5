"""Train step with the main chain on a HIGH-priority stream (the library's side stream stays at normal priority) vs the default
(both normal).  GPU box.  usage: python tools/prio_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_amd.config import Config
from tacotron_amd.data import synthetic_batch
from tacotron_amd.model import Tacotron
c = Config(); c.r, c.vocab_size = 2, 60
m = Tacotron(c, synthetic_batch(32, 200, 180, 2, 60), train=True, seed=0)


def timed(n=30):
    for _ in range(3): m.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): m.step()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


hp = torch.cuda.Stream(priority=-1)
for rep in range(3):
    a = timed()
    with torch.cuda.stream(hp):
        b = timed()
    print('default stream %.3f ms   high-priority main stream %.3f ms' % (a, b), flush=True)
m.check()
