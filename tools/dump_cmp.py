"""Compare two dumps of tools/dump_bwd.py (on the GPU box): python tools/dump_cmp.py old new"""
import sys, numpy as np
sys.path.insert(0, '.')
a, b = (np.load('/tmp/dump_%s.npz' % t) for t in sys.argv[1:3])
GS = dict(G=(0, 1536), C=(1536, 2304), X=(2304, 2560), Q=(2560, 2816), P1S=(2816, 3072), Ctx=(3072, 3328), P2=(3328, 3456), P1=(3456, 3712), O=(3712, 3872))
def rel(x, y): return float(np.linalg.norm(x - y) / max(np.linalg.norm(x), 1e-30))
for k in ('s2s', 'al', 'bwd.dkeys', 'grads'):
    print('%-12s rel %.3e' % (k, rel(a[k], b[k])))
ga, gb = a['bwd.gstash'].reshape(4, 12, -1), b['bwd.gstash'].reshape(4, 12, -1)
for n, (lo, hi) in GS.items():
    print('gstash %-4s rel %.3e   per-step:' % (n, rel(ga[..., lo:hi], gb[..., lo:hi])), ' '.join('%.1e' % rel(ga[:, t, lo:hi], gb[:, t, lo:hi]) for t in range(12)))
sa, sb = a['dec.stash'].reshape(4, 12, -1), b['dec.stash'].reshape(4, 12, -1)
ST = dict(P1=(0, 256), P2=(256, 384), X=(384, 640), H=(640, 1408), R=(1408, 2176), U=(2176, 2944), C=(2944, 3712), RH=(3712, 4480), Ctx=(4480, 4736), Q=(4992, 5248), Y=(5248, 5504))
for n, (lo, hi) in ST.items():
    print('stash %-4s rel %.3e' % (n, rel(sa[..., lo:hi], sb[..., lo:hi])))
for l in range(3):
    print('G layer %d per-step:' % l, ' '.join('%.1e' % rel(ga[:, t, l * 512:(l + 1) * 512], gb[:, t, l * 512:(l + 1) * 512]) for t in range(12)))
    print('C layer %d per-step:' % l, ' '.join('%.1e' % rel(ga[:, t, 1536 + l * 256:1536 + (l + 1) * 256], gb[:, t, 1536 + l * 256:1536 + (l + 1) * 256]) for t in range(12)))
for bb in range(4):
    print('row %d: Q step10 %.1e  G2 step10 %.1e  X step 10 %.1e' % (bb, rel(ga[bb, 10, 2560:2816], gb[bb, 10, 2560:2816]), rel(ga[bb, 10, 1024:1536], gb[bb, 10, 1024:1536]), rel(ga[bb, 10, 2304:2560], gb[bb, 10, 2304:2560])))
