"""Inference decode timing at B = 1, 8, 16 (decoder3 R = 1, 1, 2) -- A/B of decoder builds through TACO_LIB.  GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tacotron_amd import lib
from tacotron_amd.config import Config
from tacotron_amd.data import synthetic_batch
from tacotron_amd.model import Tacotron
ci = Config(); ci.r, ci.vocab_size, ci.max_decode_iter = 2, 60, 180
out = []
for Bi in (1, 8, 16):
    mi = Tacotron(ci, synthetic_batch(Bi, 140, 180, 2, 60, seed=77, min_len=40), train=False, seed=0)
    for _ in range(3): mi.run()
    torch.cuda.synchronize(); lib.profile_read(0); lib.profile_enable(1)
    its = []
    for _ in range(15):
        t0 = time.perf_counter(); mi.run(); torch.cuda.synchronize(); its.append((time.perf_counter() - t0) * 1e3)
    lib.profile_enable(0); f = lib.profile_read(0); mi.check()
    out.append('B=%d: %.3f ms/batch, decoder %.2f us/step' % (Bi, np.median(its), np.median(f) * 1e3 / 180))
print(' | '.join(out))
