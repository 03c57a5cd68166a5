"""Is the HIP-vs-fp64 error of the encoder-bottom gradients (embedding, encoder pre_net layer 1) rounding noise?
Runs the S1 step three times in the default mode (fp32 atomics decide the order of the split-M / K-way sums) and once in
TACO_DETERMINISTIC=1 mode and prints per-tensor rel-L2 differences between the runs, next to the error against the fp64 oracle
when --oracle is given.  usage: python tools/grad_noise.py [speakers] [--oracle]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tacotron_amd import lib
from tacotron_amd.data import synthetic_batch
from tests.test_gpu_model import Runner, f64
from tests.util import rel_l2
S = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1
B, Tt, Td, r, V = 32, 200, 180, 2, 60
batch = synthetic_batch(B, Tt, Td, r, V, num_speakers=S)
inp = {k: batch[k].numpy() for k in ('text', 'text_length', 'mel', 'stft') + (('speaker',) if S > 1 else ())}
rng = np.random.default_rng(1 if S > 1 else 0)
masks = {'enc_keep1': rng.integers(0, 2, (B, Tt, 256)), 'enc_keep2': rng.integers(0, 2, (B, Tt, 128)),
         'dec_keep1': rng.integers(0, 2, (B, Td, 256)), 'dec_keep2': rng.integers(0, 2, (B, Td, 128)),
         'sample': rng.integers(0, 2, (Td, B))}
R = Runner(lib, B, Tt, Td, r, V, S=S)
R.pb.init_(seed=0)
p = R.pb.to_dict()
R.set(p, inp, masks)
runs = []
for i in range(4):
    if i == 3:
        os.environ['TACO_DETERMINISTIC'] = '1'
    R.forward(); R.backward()
    runs.append(R.pb.to_dict(R.grads))
ref = None
if '--oracle' in sys.argv:
    from oracle import taco_torch as ot
    torch.set_num_threads(16)
    ref = ot.loss_and_grads(p, f64(inp), r, Td, f64(masks))[4]
names = ['embedding', 'encoder/pre_net/dense/kernel', 'encoder/pre_net/dense_1/kernel', 'encoder/cbhg/bank_1/kernel',
         'encoder/cbhg/bank_16/kernel', 'encoder/cbhg/proj1/kernel', 'encoder/cbhg/proj2/kernel', 'encoder/cbhg/highway_0/T/kernel',
         'encoder/cbhg/bigru/fw/gates/kernel', 'decoder/memory_layer/kernel', 'decoder/gru_0/gates/kernel', 'post/cbhg/proj1/kernel']
print('%-40s %10s %10s %10s %10s' % ('tensor (S=%d)' % S, 'run1-run0', 'run2-run0', 'det-run0', 'vs fp64'))
for n in names:
    print('%-40s %10.2e %10.2e %10.2e %10s' % (n, rel_l2(runs[1][n], runs[0][n]), rel_l2(runs[2][n], runs[0][n]),
                                             rel_l2(runs[3][n], runs[0][n]), '%.2e' % rel_l2(runs[0][n], ref[n]) if ref else '-'))
