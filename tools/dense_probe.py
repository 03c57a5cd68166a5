"""The final dense layer's forms in a steady loop (GPU box): aligned pitch, pitch 1025 (shifted float4 epilogue), + the column kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_amd import lib


def steady(fn, seconds=0.5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize(); n += 20
    return (time.perf_counter() - t0) / n * 1e6


def cold(fn, flush, n=6):
    ts = []
    for _ in range(n):
        flush.add_(1.0)   # 1 GB read-modify-write: evicts L2 and the Infinity Cache
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


M, K = 11520, 256
A = torch.randn(M, K, device='cuda'); Wp = torch.randn(K, 1028, device='cuda') * 0.05; bias = torch.randn(1028, device='cuda')
flush = torch.zeros(256 * 1024 * 1024, device='cuda')
for name, N, nld, ldc in [('N=1024 pitch 1024 (float4 epilogue)', 1024, 1024, 1024), ('N=1024 pitch 1025 (shifted epilogue)', 1024, 1024, 1025),
                          ('N=1024 pitch 1028 (float4 epilogue)', 1024, 1024, 1028), ('N=1025 pitch 1025 (9 tiles, shifted)', 1025, 1028, 1025)]:
    C = torch.empty(M * ldc + 64, device='cuda')
    fn = lambda: lib.conv_gemm_nld(A, Wp, C, M, N, K, nld, 1028, ldc, act=0, bias=bias)
    print('%-40s steady %6.1f us   cold-cache %6.1f us' % (name, steady(fn), cold(fn, flush)), flush=True)
B2 = Wp[:, :1024].contiguous()
fn = lambda: torch.mm(A, B2)
print('%-40s steady %6.1f us   cold-cache %6.1f us' % ('vendor BLAS N=1024', steady(fn), cold(fn, flush)), flush=True)
