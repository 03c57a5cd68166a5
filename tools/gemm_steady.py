"""Cold vs steady-state rate of the NN kernel on the step's main shapes, beside the vendor BLAS on the taps=1 ones (GPU box).
Per shape: launches timed one by one with HIP events right after an idle gap (first 8 shown), then a 1.5 s steady loop.
usage: python tools/gemm_steady.py > gpurun_out/r04_gemm_steady.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_amd import lib

SHAPES = [   # name, M, N, K, taps, T
    ('square 4096^3', 4096, 4096, 4096, 1, 4096),
    ('enc proj1  M=6400 N=128 K=2048 taps=3', 6400, 128, 2048, 3, 200),
    ('post proj1 M=11520 N=256 K=1024 taps=3', 11520, 256, 1024, 3, 360),
    ('post proj1 dX M=11520 N=1024 K=256 taps=3', 11520, 1024, 256, 3, 360),
    ('final dense M=11520 N=1025 K=256', 11520, 1025, 256, 1, 11520),
    ('final dense dX M=11520 N=256 K=1025', 11520, 256, 1025, 1, 11520),
    ('bank tap-8 member M=6400 N=128 K=128 taps=8', 6400, 128, 128, 8, 200),
    ('x-projection M=11520 N=768 K=128', 11520, 768, 128, 1, 11520),
]


def timed(fn, n):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]


def steady(fn, seconds=1.5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize(); n += 20
    dt = time.perf_counter() - t0
    last = timed(fn, 20)
    return dt / n * 1e3, sum(last) / len(last)


def main():
    for name, M, N, K, taps, T in SHAPES:
        A = torch.randn(M, K, device='cuda')
        W = torch.randn(taps, K, N, device='cuda') * 0.05
        C = torch.empty(M, N, device='cuda')
        gf = 2.0 * M * N * K * taps / 1e9
        ours = lambda: lib.conv_gemm(A, W, C, M, N, K, taps=taps, T=T, pad_l=(taps - 1) // 2, act=0)
        ours(); torch.cuda.synchronize()
        time.sleep(0.5)
        cold = timed(ours, 8)
        wall, ev = steady(ours)
        print('%-46s %6.2f GF | cold launches (us): %s | steady %.1f us = %.1f TF (events %.1f us)' % (
            name, gf, ' '.join('%.0f' % (x * 1e3) for x in cold), wall * 1e3, gf / wall, ev * 1e3), flush=True)
        if taps == 1:
            Wm = W[0]
            blas = lambda: torch.mm(A, Wm, out=C)
            blas(); torch.cuda.synchronize()
            time.sleep(0.5)
            cold = timed(blas, 8)
            wall, ev = steady(blas)
            print('%-46s %6s    | cold launches (us): %s | steady %.1f us = %.1f TF (events %.1f us)' % (
                '   vendor BLAS (torch.mm)', '', ' '.join('%.0f' % (x * 1e3) for x in cold), wall * 1e3, gf / wall, ev * 1e3), flush=True)


if __name__ == '__main__':
    main()
