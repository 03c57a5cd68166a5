"""Sensitivity of the NN kernel to the final dense layer's awkward extents (N = 1025 columns, row pitch 1025 floats) (GPU box).
usage: python tools/gemm_shapes.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_amd import lib


def steady(fn, seconds=0.6):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize(); n += 20
    return (time.perf_counter() - t0) / n * 1e6


def run(name, M, N, K, lda=None, ldw=None, ldc=None, blas=False):
    lda, ldw, ldc = lda or K, ldw or N, ldc or N
    A = torch.randn(M, lda, device='cuda'); W = torch.randn(K, ldw, device='cuda') * 0.05; C = torch.empty(M, ldc, device='cuda')
    us = steady(lambda: lib.conv_gemm(A, W, C, M, N, K, taps=1, T=M, pad_l=0, act=0, lda=lda, ldw=ldw, ldc=ldc))
    gf = 2.0 * M * N * K / 1e9
    line = '%-58s %7.1f us %6.1f TF' % (name, us, gf / us * 1e3)
    if blas:
        Av, Wv = A[:, :K], W[:, :N]
        ub = steady(lambda: torch.mm(Av, Wv))
        line += '   | vendor BLAS %7.1f us %6.1f TF' % (ub, gf / ub * 1e3)
    print(line, flush=True)


M = 11520
run('fwd N=1024 K=256 (aligned everything)', M, 1024, 256, blas=True)
run('fwd N=1025 K=256 ldc=ldw=1025 (the model shape)', M, 1025, 256, blas=True)
run('fwd N=1025 K=256 ldc=ldw=1028', M, 1025, 256, ldw=1028, ldc=1028, blas=True)
run('fwd N=1152 K=256 (9 full n-tiles)', M, 1152, 256, blas=True)
run('dX  N=256 K=1024 (aligned everything)', M, 256, 1024, blas=True)
run('dX  N=256 K=1025 lda=1025', M, 256, 1025, blas=True)
run('dX  N=256 K=1025 lda=1028 (the model shape)', M, 256, 1025, lda=1028, blas=True)
run('fwd N=1024 K=512', M, 1024, 512, blas=True)
run('fwd N=1024 K=1024', M, 1024, 1024, blas=True)
run('M=6400 N=128 K=2048 (enc proj1 as a plain GEMM)', 6400, 128, 2048, blas=True)
run('M=6400 N=2048 K=128', 6400, 2048, 128, blas=True)
run('M=11520 N=128 K=768 (gh)', M, 128, 768, blas=True)
