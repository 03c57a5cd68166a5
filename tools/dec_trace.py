"""Prints the per-phase time split of the decoder forward kernel (block 0, step Td/2) recorded with TACO_DEC_TRACE=1.
usage (GPU box): TACO_DEC_TRACE=1 python tools/dec_trace.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['TACO_DEC_TRACE'] = '1'
import torch
from tacotron_amd.config import Config
from tacotron_amd.data import synthetic_batch
from tacotron_amd.model import Tacotron
from tacotron_amd import lib
c = Config(); c.r, c.vocab_size = 2, 60
m = Tacotron(c, synthetic_batch(32, 200, 180, 2, 60), train=True, seed=0)
masks = m.draw_masks()
for _ in range(3):
    m.forward(masks)
torch.cuda.synchronize()
tab = {n: (o, s) for n, o, s, d in lib.workspace_table(m.shape, True)}
o, s = tab['dec.err']
tr = m.workspace[o + 16:o + 16 + 4 * 2 * 20].view(torch.int64).cpu().numpy().reshape(-1, 4)
names = ['p1', 'p2', 'x', 'g0', 'c0', 'g1', 'c1', 'g2', 'c2', 'out', 'q', 'e', 'ctx', 'att']
t0 = tr[0, 0]
print('phase   start_us  matvec  barrier+finalize  gather   total   (wall_clock64 = 100 MHz ticks)')
for i, n in enumerate(names):
    a, b, c_, d = [(x - t0) / 100.0 for x in tr[i]]
    nxt = (tr[i + 1, 0] - t0) / 100.0 if i + 1 < len(names) else float('nan')
    print('%-5s %9.2f %8.2f %12.2f %10.2f %8.2f' % (n, a, b - a, c_ - b, d - c_, nxt - a))
print('step total us: %.2f' % ((tr[len(names) - 1, 3] - t0) / 100.0))
