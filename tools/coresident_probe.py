"""Does GEMM work co-reside with the persistent decoder kernel (1 workgroup per CU, 8 waves, ~170 VGPRs, 54 KB LDS) and what
does each side pay?  Stream A: train forward (decoder forward kernel timed by the library's event ring).  Stream B: a loop of
weight-gradient GEMMs (post-net conv-bank dW shape).  usage: python tools/coresident_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_amd import lib
from tacotron_amd.config import Config
from tacotron_amd.data import synthetic_batch
from tacotron_amd.model import Tacotron
c = Config(); c.r, c.vocab_size = 2, 60
m = Tacotron(c, synthetic_batch(32, 200, 180, 2, 60), train=True, seed=0)
masks = m.draw_masks()
M, T, N, K, taps = 11520, 360, 128, 80, 8
A = torch.randn(M, K, device='cuda'); Y = torch.randn(M, N, device='cuda'); dW = torch.zeros(taps, K, N, device='cuda')
M2, N2, K2 = 11520, 256, 1024
A2 = torch.randn(M2, K2, device='cuda'); Y2 = torch.randn(M2, N2, device='cuda'); dW2 = torch.zeros(3, K2, N2, device='cuda')
sb = torch.cuda.Stream()
def gemms(n):
    for _ in range(n):
        lib.gemm_tn(A, Y, dW, M, N, K, taps=taps, T=T, pad_l=3, accumulate=True)
        lib.gemm_tn(A2, Y2, dW2, M2, N2, K2, taps=3, T=T, pad_l=1, accumulate=True)
for _ in range(2): m.forward(masks); gemms(2)
torch.cuda.synchronize()
# GEMMs alone
t0 = time.perf_counter(); gemms(20); torch.cuda.synchronize(); g_alone = (time.perf_counter() - t0) / 20 * 1e3
# forward alone
lib.profile_read(0); lib.profile_enable(1)
for _ in range(5): m.forward(masks)
torch.cuda.synchronize(); f_alone = lib.profile_read(0)
# together: GEMM loop on stream B while the forward runs on the current stream
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(sb):
    ev0.record()
    gemms(60)
    ev1.record()
for _ in range(5): m.forward(masks)
torch.cuda.synchronize()
f_both = lib.profile_read(0); lib.profile_enable(0)
print('GEMM pair alone: %.3f ms; 60 pairs concurrently with 5 forwards: %.3f ms per pair' % (g_alone, ev0.elapsed_time(ev1) / 60))
print('decoder fwd alone: %s ms' % ['%.2f' % x for x in f_alone])
print('decoder fwd with GEMMs on another stream: %s ms' % ['%.2f' % x for x in f_both])
