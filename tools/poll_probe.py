"""Decoder exchange: single poll vs pipelined polls (TACO_DEC_POLL="<extra>,<gap>"), GPU box.  Results stay correct."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_amd import lib
from tacotron_amd.config import Config
from tacotron_amd.data import synthetic_batch
from tacotron_amd.model import Tacotron
c = Config(); c.r, c.vocab_size = 2, 60
m = Tacotron(c, synthetic_batch(32, 200, 180, 2, 60), train=True, seed=0)
modes = sys.argv[1:] or ['', '1,0', '1,1', '1,2', '1,4', '2,1', '2,2', '2,4', '']
for mode in modes:
    os.environ.pop('TACO_DEC_POLL', None)
    if mode: os.environ['TACO_DEC_POLL'] = mode
    masks = m.draw_masks()
    for _ in range(2): m.forward(masks); m.backward()
    torch.cuda.synchronize()
    lib.profile_read(0); lib.profile_read(1); lib.profile_enable(3)
    for _ in range(6): m.forward(masks); m.backward()
    torch.cuda.synchronize(); lib.profile_enable(0)
    f, b = lib.profile_read(0), lib.profile_read(1)
    print('poll %-5s fwd %.3f ms (%.2f us/step)  bwd %.3f ms (%.2f us/step)  loss %.6g err %s' % (mode or 'single', sum(f) / len(f), sum(f) / len(f) / 180 * 1e3, sum(b) / len(b), sum(b) / len(b) / 180 * 1e3, float(m.loss), m._err.tolist()[:2]), flush=True)
