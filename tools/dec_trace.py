"""Prints the per-phase time split of decoder.hip's kernels (block 0, step Td/2) recorded with TACO_DEC_TRACE=1.  These are the
round-1/2 kernels (TACO_DEC_V3=0 is forced here); decoder3.hip has its own trace: tools/dec3_trace.py.  Rounds that do not stamp
every one of their four slots (E has no mid-round barrier: no stamp 1; the backward FAN round only stamps 1 / 2 when the pre-net
rider runs) are printed with '-' for the missing split instead of the garbage an unstamped (zero) slot used to produce.
usage (GPU box): TACO_DEC_TRACE=1 python tools/dec_trace.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['TACO_DEC_TRACE'] = '1'
os.environ['TACO_DEC_V3'] = '0'
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
names = ['g0+x', 'c0', 'g1', 'c1', 'g2', 'c2', 'out+q+p1n', 'e+p2n']
t0 = tr[0, 0]
print('phase   start_us  matvec  barrier+finalize  gather   total   (wall_clock64 = 100 MHz ticks)')
def row(n, st, nxt0):
    """st: the four stamps of a round (0 = not stamped).  Splits are only printed between stamps that exist."""
    us = [(x - t0) / 100.0 if x else None for x in st]
    def d(i, j):
        return '%8.2f' % (us[j] - us[i]) if us[i] is not None and us[j] is not None else '       -'
    tot = '%8.2f' % ((nxt0 - t0) / 100.0 - us[0]) if nxt0 and us[0] is not None else '       -'
    print('%-12s %9s %s %s %s %s' % (n, '%.2f' % us[0] if us[0] is not None else '-', d(0, 1), d(1 if us[1] is not None else 0, 2), d(2, 3), tot))
for i, n in enumerate(names):
    row(n, tr[i], tr[i + 1, 0] if i + 1 < len(names) else 0)
print('step total us: %.2f' % ((tr[len(names) - 1, 3] - t0) / 100.0))
mk = m.workspace[o + 16 + 320:o + 16 + 320 + 2 * 32].view(torch.int64).cpu().numpy()
print('fwd attention marks (us since e-phase start):', ['%.2f' % ((mk[i] - mk[0]) / 100.0) for i in range(6)],
      ' [0 start,1 energies done,2 gathered,3 barrier,4 softmax done,5 barrier]')
ck = m.workspace[o + 16 + 380:o + 16 + 380 + 4].view(torch.int64).cpu().numpy()
print('shader clock during the decoder kernel: %.0f MHz' % ((ck[1] - ck[0]) / ((mk[5] - mk[0]) / 100.0)))
# ---- backward ----
m.backward()
torch.cuda.synchronize()
tb = m.workspace[o + 16 + 256:o + 16 + 256 + 4 * 2 * 20].view(torch.int64).cpu().numpy().reshape(-1, 4)
bn = ['fan+dal+dp2', 'outT', 'c2T', 'g2T', 'c1T', 'g1T', 'c0T', 'g0T']
t0 = tb[0, 0]
print('BACKWARD phase start_us matvec finalize gather total')
for i, n in enumerate(bn):
    if tb[i, 0] == 0: break
    row(n, tb[i], tb[i + 1, 0] if i + 1 < len(bn) else 0)

mk = m.workspace[o + 16 + 256 + 320:o + 16 + 256 + 320 + 2 * 32].view(torch.int64).cpu().numpy()
print('bwd marks (us since FAN start):', ['%.2f' % ((mk[i] - mk[10]) / 100.0) for i in range(10, 19)],
      ' [10 start,11 dal rows done,12 fan gathered,13 barrier,14 softmax-bwd done,15 energy-bwd + dp1 mat-vec done,16 barrier,17 dq/dp1 exchanged,18 barrier]')
