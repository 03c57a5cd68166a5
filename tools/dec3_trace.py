"""Per-round time split of decoder3_fwd_kernel (probe build, workgroup 0, step Td/2): shader-clock stamps at
  round start -> computed+published -> gathered -> barrier.   usage (GPU box): python tools/dec3_trace.py [B] [infer]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['TACO_DEC_TRACE'] = '1'
os.environ.setdefault('TACO_LIB', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tacotron_amd', 'libtaco_probe.so'))
import torch
from tacotron_amd.config import Config
from tacotron_amd.data import synthetic_batch
from tacotron_amd.model import Tacotron
from tacotron_amd import lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
infer = len(sys.argv) > 2
c = Config(); c.r, c.vocab_size, c.max_decode_iter = 2, 60, 180
m = Tacotron(c, synthetic_batch(B, 200, 180, 2, 60), train=not infer, seed=0)
if infer:
    for _ in range(3): m.run()
else:
    masks = m.draw_masks()
    for _ in range(3): m.forward(masks)
torch.cuda.synchronize()
tab = {n: (o, s) for n, o, s, d in lib.workspace_table(m.shape, not infer)}
o, s = tab['dec.err']
tr = m.workspace[o + 16:o + 16 + 2 * 64].view(torch.int64).cpu().numpy().copy()
n_ = int(tr[63]); pol = [(int(x) >> 56) & 0xff for x in tr[:n_]]; tr[:n_] &= 0x00ffffffffffffff
n = int(tr[63]); polls = int(tr[62]); print('poll iterations of the E gather (max over the traced workgroup, summed over launches):', int(tr[61]) & 0xffffffff)
t = [(x - tr[0]) / 2400.0 for x in tr[:n]]      # shader clock ~2.4 GHz (reported in us at that nominal rate)
names = ['G0', 'C0', 'G1', 'C1', 'G2', 'C2', 'OUT']
print('B=%d %s: %d stamps, max poll iterations of a thread in the step: %d' % (B, 'infer' if infer else 'train', n, polls))
print('round   compute+publish   gather   barrier   (us at 2.4 GHz)')
i = 1
tot = [0, 0, 0]
for nm in names:
    a, b, c_ = t[i] - t[i - 1], t[i + 1] - t[i], t[i + 2] - t[i + 1]
    print('%-5s %12.2f %12.2f %9.2f' % (nm, a, b, c_))
    tot = [tot[0] + a, tot[1] + b, tot[2] + c_]
    i += 3
rest = t[i:]
print('E / softmax stamps since the OUT barrier [energies published, p2 rider done, gathered, deferred stores + next-step loads issued, barrier, softmax + barrier]:', ['%.2f' % (x - t[i - 1]) for x in rest])
print('sum over G0..OUT: compute %.2f gather %.2f barrier %.2f; whole step %.2f us' % (tot[0], tot[1], tot[2], t[n - 1]))
print('poll-loop passes of wave 0 at each stamp (cumulative; a gather that hits on its first check adds 1):', pol)

if not infer:
    m.backward()
    torch.cuda.synchronize()
    tb = m.workspace[o + 16 + 256:o + 16 + 256 + 2 * 64].view(torch.int64).cpu().numpy().copy()
    nb = int(tb[63]); polb = [(int(x) >> 56) & 0xff for x in tb[:nb]]; tb[:nb] &= 0x00ffffffffffffff
    tt = [(x - tb[0]) / 2400.0 for x in tb[:nb]]
    names = ['inputs landed', 'FAN computed + published', 'FAN gathered + barrier', 'softmax backward + barrier', 'DQ energy backward', 'DQ dq + d p1 published',
             'DQ gathered', 'DQ stores + barrier', 'OUT computed + published', 'OUT gathered', 'OUT barrier']
    for l in (2, 1, 0):
        names += ['C%d computed + published' % l, 'C%d gathered' % l, 'C%d store + barrier' % l, 'G%d computed + published' % l, 'G%d gathered' % l, 'G%d barrier' % l]
    print('BACKWARD step (us at 2.4 GHz), section durations:')
    for i in range(1, nb):
        print('  %-36s %6.2f' % (names[i - 1] if i - 1 < len(names) else '?', tt[i] - tt[i - 1]))
    print('  whole step %.2f us' % tt[nb - 1])
    print('  poll-loop passes of wave 0 at each stamp (cumulative):', polb)
