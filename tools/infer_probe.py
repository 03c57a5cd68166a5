"""Times taco_infer at B=1 / B=32 (GPU box).  usage: python tools/infer_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_amd.config import Config
from tacotron_amd.data import synthetic_batch
from tacotron_amd.model import Tacotron
c = Config(); c.r, c.vocab_size, c.max_decode_iter = 2, 60, 180
for Bi in (1, 32):
    b = synthetic_batch(Bi, 140, 180, 2, 60, seed=77, min_len=40)
    m = Tacotron(c, b, train=False, seed=0)
    for _ in range(2): m.run()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): m.run()
    torch.cuda.synchronize()
    print('B=%d %.2f ms  env cluster=%s overlap_off=%s' % (Bi, (time.perf_counter() - t) / 5 * 1e3, os.environ.get('TACO_DEC_CLUSTER'), os.environ.get('TACO_NO_OVERLAP')), flush=True)
    m.check()
