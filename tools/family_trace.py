"""Per-launch HIP-event times of the MFMA GEMM family over one train step (GPU box).  usage: python tools/family_trace.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_amd import lib
from tacotron_amd.config import Config
from tacotron_amd.data import synthetic_batch
from tacotron_amd.model import Tacotron
c = Config(); c.r, c.vocab_size = 2, 60
m = Tacotron(c, synthetic_batch(32, 200, 180, 2, 60), train=True, seed=0)
for _ in range(3): m.step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): m.step()
e1.record(); torch.cuda.synchronize()
print('step %.3f ms' % (e0.elapsed_time(e1) / 10))
lib.profile_read(2); lib.profile_read(3)
lib.profile_enable(0b11100)   # bit 4: side stream off, every launch timed by itself
N = 4
for _ in range(N): m.step()
torch.cuda.synchronize()
lib.profile_enable(0)
labels = lib.profile_labels(2)
ms, fl = lib.profile_read(2, with_flops=True)
n = len(ms) // N
tot = 0.0
print('%d launches per step' % n)
# rates are fp32-EQUIVALENT TFLOP/s.  Launches of the bf16x3 form (six v_mfma_f32_32x32x16_bf16 per fp32 product) run under
# 2516 / 6 = 419 TF, not under the 157.3 TF fp32-instruction peak: a rate above 157.3 is marked '>fp32pk' and priced against 419
print('   (TFe = fp32-equivalent TFLOP/s; %%own = of the 419 TFe bf16x3 roof for rates above the 157.3 TF fp32-instruction peak, else of 157.3)')
for i in range(n):
    t = sum(ms[i + k * n] for k in range(N)) / N
    tot += t
    tf = fl[i] / (t * 1e-3) / 1e12 if t > 0 else 0
    own = '%4.1f%% of 419 >fp32pk' % (100 * tf / 419.3) if tf > 157.3 else '%4.1f%% of 157.3' % (100 * tf / 157.3)
    print('#%2d %8.1f us %8.2f GF %6.1f TFe (%s)  %s' % (i, t * 1e3, fl[i] / 1e9, tf, own, labels[i] if i < len(labels) else ''))
print('sum %.1f us' % (tot * 1e3))
rm = lib.profile_read(3)
print('bigru', ['%.1f' % (x * 1e3) for x in rm[:4]])
