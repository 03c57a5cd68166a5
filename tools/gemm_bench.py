"""Times taco_conv_gemm / taco_gemm_tn at the model's real shapes (GPU box).  usage: python tools/gemm_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_amd import lib

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3  # us

NN = [  # name, M, T, N, K, taps
    ('enc proj1 fwd', 6400, 200, 128, 2048, 3),
    ('post proj1 fwd', 11520, 360, 256, 1024, 3),
    ('enc dpool (bwd)', 6400, 200, 2048, 128, 3),
    ('post dpool (bwd)', 11520, 360, 1024, 256, 3),
    ('post dense fwd', 11520, 11520, 1025, 256, 1),
    ('post dense bwd', 11520, 11520, 256, 1028, 1),
    ('bank k=16 alone', 6400, 200, 128, 128, 16),
    ('highway 128', 6400, 6400, 128, 128, 1),
    ('enc prenet', 6400, 6400, 256, 256, 1),
    ('square 4096', 4096, 4096, 4096, 4096, 1),
    ('enc bank k=8 alone', 6400, 200, 128, 128, 8),
]
for name, M, T, N, K, taps in NN:
    A = torch.randn(M, K, device='cuda'); W = torch.randn(taps, K, N, device='cuda') * 0.05; C = torch.empty(M, N, device='cuda')
    us = timeit(lambda: lib.conv_gemm(A, W, C, M, N, K, taps=taps, T=T, pad_l=(taps - 1) // 2, act=1))
    gf = 2.0 * M * N * K * taps / 1e9
    print('NN %-18s M=%5d N=%4d K=%4d taps=%2d  %8.1f us  %6.1f TF' % (name, M, N, K, taps, us, gf / us * 1e-3 * 1e3 / 1e3 * 1e3 / 1e3 if False else gf / (us * 1e-6) / 1e3))
TN = [
    ('enc proj1 dW', 6400, 200, 128, 2048, 3),
    ('post proj1 dW', 11520, 360, 256, 1024, 3),
    ('post dense dW', 11520, 11520, 1025, 256, 1),
    ('dec gru gates dW', 5760, 180, 512, 256, 1),
    ('bank k=16 dW', 6400, 200, 128, 128, 16),
    ('highway dW', 6400, 6400, 128, 128, 1),
]
for name, M, T, N, K, taps in TN:
    A = torch.randn(M, K, device='cuda'); Y = torch.randn(M, N, device='cuda'); dW = torch.zeros(taps, K, N, device='cuda')
    us = timeit(lambda: lib.gemm_tn(A, Y, dW, M, N, K, taps=taps, T=T, pad_l=(taps - 1) // 2, accumulate=True))
    gf = 2.0 * M * N * K * taps / 1e9
    print('TN %-18s M=%5d N=%4d K=%4d taps=%2d  %8.1f us  %6.1f TF' % (name, M, N, K, taps, us, gf / (us * 1e-6) / 1e3))
