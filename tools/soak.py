"""Soak run (GPU box): N train steps on one synthetic batch at the benchmark shape; prints the loss trajectory, the decoder error
words and the placement census.  usage: python tools/soak.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_amd.config import Config
from tacotron_amd.data import synthetic_batch
from tacotron_amd.model import Tacotron
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
c = Config(); c.r, c.vocab_size = 2, 60
m = Tacotron(c, synthetic_batch(32, 200, 180, 2, 60), train=True, seed=0)
t0 = time.perf_counter()
for i in range(n):
    m.step()
    if i % (n // 10) == 0 or i == n - 1:
        m.check()
        print('step %5d  loss %.6g  grad-norm %.4g' % (i, float(m.loss), float(m.global_gradient_norm)), flush=True)
torch.cuda.synchronize()
print('%d steps, %.2f ms/step incl. host checks; error words %s; census %s' % (n, (time.perf_counter() - t0) / n * 1e3, m._err.tolist()[:2], m.placement_census()))
