"""Stage stamps of the fused pre_net kernel.  Needs the probe build: `bash tools/ab_build.sh pntrace "-DTACO_PN_TRACE" prenet`, then
  TACO_LIB=$PWD/tacotron_amd/libtaco_pntrace.so python tools/pn_trace.py   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['TACO_PN_TRACE'] = '1'
import torch
from tacotron_amd.config import Config
from tacotron_amd.data import synthetic_batch
from tacotron_amd.model import Tacotron
from tacotron_amd import lib
c = Config(); c.r, c.vocab_size, c.max_decode_iter = 2, 60, 180
m = Tacotron(c, synthetic_batch(32, 200, 180, 2, 60), train=True, seed=0)
masks = m.draw_masks()
for _ in range(3): m.forward(masks)
torch.cuda.synchronize()
tab = {n: (o, s) for n, o, s, d in lib.workspace_table(m.shape, True)}
o, s = tab['dec.err']
tr = m.workspace[o + 400:o + 400 + 16].view(torch.int64).cpu().numpy()
names = ['input tile -> LDS (+ first W chunk issued)', 'stage 1 MFMAs', 'epilogue 1', 'barrier', 'stage 2 MFMAs', 'epilogue 2']
for i in range(6):
    print('%-44s %7.2f us' % (names[i], (tr[i + 1] - tr[i]) / 2400.0))   # shader clock, ~2.4 GHz nominal
print('whole workgroup %.2f us' % ((tr[6] - tr[0]) / 2400.0))
