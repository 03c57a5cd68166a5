"""Soak of the other measured configurations (GPU box): S2 (Td = 500), B = 64 per GPU, VCTK-shaped (109 speakers), r = 5 -- N train steps each
on one synthetic batch; prints ms/step, the decoder error words and whether the loss stayed finite.  usage: python tools/soak_shapes.py [steps]"""
import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_amd.config import Config
from tacotron_amd.data import synthetic_batch
from tacotron_amd.model import Tacotron
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
for name, B, Tt, Td, r, S in (('S2 Td=500', 32, 200, 500, 2, 1), ('B=64', 64, 200, 180, 2, 1), ('VCTK 109 speakers', 32, 200, 180, 2, 109),
                              ('r=5 Td=72', 32, 200, 72, 5, 1), ('B=50 ragged clusters', 50, 200, 180, 2, 1)):
    c = Config(); c.r, c.vocab_size, c.num_speakers = r, 60, S
    m = Tacotron(c, synthetic_batch(B, Tt, Td, r, 60, num_speakers=S), train=True, seed=0)
    for _ in range(3): m.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    l0 = None
    for i in range(n):
        m.step()
        if i == 0: l0 = float(m.loss)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    m.check()
    l1 = float(m.loss)
    print('%-22s %4d steps  %.2f ms/step  loss %.5g -> %.5g (finite: %s)  error words %s  decoder mode %d' %
          (name, n, dt * 1e3, l0, l1, math.isfinite(l1), m._err.tolist()[:2], __import__('tacotron_amd').lib.decoder_mode()), flush=True)
    del m
    torch.cuda.empty_cache()
