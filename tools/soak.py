import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from tacotron_amd.config import Config
from tacotron_amd.data import synthetic_batch
from tacotron_amd.model import Tacotron
c = Config(); c.r, c.vocab_size = 2, 60
b = synthetic_batch(32, 200, 180, 2, 60, seed=5)
m = Tacotron(c, b, train=True, seed=1)
t0 = time.time()
for i in range(400):
    m.step(5e-4)
    if i % 100 == 99:
        torch.cuda.synchronize()
        print(i + 1, float(m.loss), float(m.global_gradient_norm), flush=True)
torch.cuda.synchronize()
print('400 steps in %.1f s' % (time.time() - t0))
