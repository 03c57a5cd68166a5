"""Where does the HIP-vs-fp64 deviation of the encoder-bottom gradients live?  The embedding table is replaced by one row per
text POSITION (V = B*Tt, ids = arange), so d loss / d embedding is the per-position gradient of the encoder input.  A deviation
caused by a handful of discrete decisions (ReLU / max-pool / dropout ties inside the CBHG, decided differently in fp32 and
fp64) is confined to the receptive field of those decisions (a few dozen positions of one sequence); accumulated rounding
would be spread over all positions.  usage: python tools/grad_localize.py [speakers]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tacotron_amd import lib
from tacotron_amd.data import synthetic_batch
from oracle import taco_torch as ot
from tests.test_gpu_model import Runner, f64
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B, Tt, Td, r, V0 = 32, 200, 180, 2, 60
batch = synthetic_batch(B, Tt, Td, r, V0, num_speakers=S)
inp = {k: batch[k].numpy() for k in ('text', 'text_length', 'mel', 'stft') + (('speaker',) if S > 1 else ())}
rng = np.random.default_rng(1 if S > 1 else 0)
masks = {'enc_keep1': rng.integers(0, 2, (B, Tt, 256)), 'enc_keep2': rng.integers(0, 2, (B, Tt, 128)),
         'dec_keep1': rng.integers(0, 2, (B, Td, 256)), 'dec_keep2': rng.integers(0, 2, (B, Td, 128)),
         'sample': rng.integers(0, 2, (Td, B))}
R0 = Runner(lib, B, Tt, Td, r, V0, S=S)
R0.pb.init_(seed=0)
p0 = R0.pb.to_dict()
del R0
V = B * Tt
p = dict(p0)
p['embedding'] = p0['embedding'][inp['text']].reshape(V, 256).copy()
inp['text'] = np.arange(V, dtype=np.int32).reshape(B, Tt)
R = Runner(lib, B, Tt, Td, r, V, S=S)
R.set(p, inp, masks)
R.forward(); R.backward()
got = R.pb.to_dict(R.grads)['embedding'].reshape(B, Tt, 256).astype(np.float64)
torch.set_num_threads(16)
ref = ot.loss_and_grads(p, f64(inp), r, Td, f64(masks))[4]['embedding'].reshape(B, Tt, 256)
err2 = ((got - ref) ** 2).sum(-1)           # (B, Tt) squared error per position
tot = err2.sum()
print('S=%d: per-position input gradient, rel-L2 overall %.3e' % (S, np.sqrt(tot) / np.linalg.norm(ref)))
flat = np.sort(err2.reshape(-1))[::-1]
for n in (1, 4, 16, 64, 256):
    print('  top %4d of %d positions carry %.1f%% of the squared error' % (n, V, 100 * flat[:n].sum() / tot))
med = np.sqrt(np.median(err2)) / np.sqrt(np.median((ref ** 2).sum(-1)))
print('  median per-position relative error %.2e' % med)
order = np.argsort(err2.reshape(-1))[::-1][:12]
print('  worst positions (b, t, text_length[b], rel err of that position):',
      [(int(i // Tt), int(i % Tt), int(inp['text_length'][i // Tt]), float('%.2e' % (np.sqrt(err2.reshape(-1)[i]) / np.linalg.norm(ref.reshape(V, 256)[i])))) for i in order])
