"""Quick decoder check: the golden r=2 fixture forward vs the committed oracle vectors, then S1 timing of the two decoder
kernels (HIP-event rings).  usage: python tools/dec_quick.py [--time-only]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tacotron_amd import lib
if '--time-only' not in sys.argv:   # (the fixture check needs the test helpers, which import the oracle; the timing path -- what bench.py's
    from tests.test_gpu_model import Runner, golden   # `ab.decoder` runs -- imports nothing outside tacotron_amd/)
    from tests.util import report
    for r in (2, 5):
        g, p, inp, masks = golden(r)
        R = Runner(lib, int(g['B']), int(g['Tt']), int(g['Td']), r, int(g['V']))
        R.set(p, inp, masks)
        R.forward()
        report('r=%d seq2seq_output' % r, R.s2s.cpu().numpy(), g['seq2seq_output'])
        report('r=%d output' % r, R.out.cpu().numpy(), g['output'])
        report('r=%d alignments' % r, R.al.cpu().numpy(), g['alignments'])
        Ri = Runner(lib, int(g['B']), int(g['Tt']), int(g['Td']), r, int(g['V']), train=False)
        Ri.set(p, {'text': inp['text'], 'text_length': inp['text_length']})
        Ri.infer()
        report('r=%d infer seq2seq_output' % r, Ri.s2s.cpu().numpy(), g['infer_seq2seq_output'])
        print('  cluster width', lib.last_cluster(0))
from tacotron_amd.config import Config
from tacotron_amd.data import synthetic_batch
from tacotron_amd.model import Tacotron
if '--json' not in sys.argv:
    print('box: %.2f GHz under a latency-bound load' % lib.clock_probe())
c = Config(); c.r, c.vocab_size = 2, 60
m = Tacotron(c, synthetic_batch(32, 200, 180, 2, 60), train=True, seed=0)
for _ in range(3): m.step()
torch.cuda.synchronize(); m.check()
lib.profile_read(0); lib.profile_read(1); lib.profile_enable(3)
import time
t0 = time.perf_counter()
for _ in range(10): m.step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
lib.profile_enable(0)
f, b = lib.profile_read(0), lib.profile_read(1)
m.check()
if '--json' in sys.argv:   # bench.py's in-run A/B of decoder builds (TACO_LIB selects the library)
    import json
    print(json.dumps({'ms_per_step': dt * 1e3, 'us_per_decoder_step_fwd': float(np.median(f)) * 1e3 / 180,
                      'us_per_decoder_step_bwd': float(np.median(b)) * 1e3 / 180, 'loss': float(m.loss)}))
    sys.exit(0)
print('S1: %.2f ms/step; decoder fwd %.3f ms (%.2f us/step, cluster %d), bwd %.3f ms (%.2f us/step); loss %.1f' %
      (dt * 1e3, np.median(f), np.median(f) * 1e3 / 180, lib.last_cluster(0), np.median(b), np.median(b) * 1e3 / 180, float(m.loss)))
ci = Config(); ci.r, ci.vocab_size, ci.max_decode_iter = 2, 60, 180
for Bi in (1, 32):
    mi = Tacotron(ci, synthetic_batch(Bi, 140, 180, 2, 60, seed=77, min_len=40), train=False, seed=0)
    for _ in range(2): mi.run()
    torch.cuda.synchronize(); lib.profile_read(0); lib.profile_enable(1)
    its = []
    for _ in range(12):
        t0 = time.perf_counter(); mi.run(); torch.cuda.synchronize(); its.append((time.perf_counter() - t0) * 1e3)
    dt = np.median(its) / 1e3
    lib.profile_enable(0); f = lib.profile_read(0); mi.check()
    print('inference B=%d: %.2f ms per batch (per run: %s); decoder %.3f ms (%.2f us/step, cluster %d)' %
          (Bi, dt * 1e3, ' '.join('%.1f' % x for x in its), np.median(f), np.median(f) * 1e3 / 180, lib.last_cluster(0)))
