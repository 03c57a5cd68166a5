"""Decoder kernel timing probes (GPU box): normal vs TACO_DEC_FAKEW=1 (all weight rows alias row 0 -> L1 hits; garbage results)."""
import os, sys
os.environ.setdefault('TACO_LIB', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tacotron_amd', 'libtaco_probe.so'))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_amd import lib
from tacotron_amd.config import Config
from tacotron_amd.data import synthetic_batch
from tacotron_amd.model import Tacotron
c = Config(); c.r, c.vocab_size = 2, 60
m = Tacotron(c, synthetic_batch(32, 200, 180, 2, 60), train=True, seed=0)
for mode in ('normal', 'fakew', 'fakex', 'fakew+x', 'nopf', 'nolive', 'nopf+nolive', 'nopf+nolive+x', 'normal'):
    for k in ('TACO_DEC_FAKEW', 'TACO_DEC_FAKEX', 'TACO_DEC_NOPF', 'TACO_DEC_NOLIVE'): os.environ.pop(k, None)
    if 'nopf' in mode: os.environ['TACO_DEC_NOPF'] = '1'
    if 'nolive' in mode: os.environ['TACO_DEC_NOLIVE'] = '1'
    if 'fakew' in mode: os.environ['TACO_DEC_FAKEW'] = '1'
    if mode.endswith('x'): os.environ['TACO_DEC_FAKEX'] = '1'
    masks = m.draw_masks()
    for _ in range(2): m.forward(masks); m.backward()
    torch.cuda.synchronize()
    lib.profile_read(0); lib.profile_read(1); lib.profile_enable(3)
    for _ in range(5): m.forward(masks); m.backward()
    torch.cuda.synchronize(); lib.profile_enable(0)
    f, b = lib.profile_read(0), lib.profile_read(1)
    print('%-13s fwd %.3f ms (%.1f us/step)  bwd %.3f ms (%.1f us/step)  err %s' % (mode, sum(f) / len(f), sum(f) / len(f) / 180 * 1e3, sum(b) / len(b), sum(b) / len(b) / 180 * 1e3, m._err.tolist()), flush=True)
    lib.clear_error(m.shape, True, m.workspace)
