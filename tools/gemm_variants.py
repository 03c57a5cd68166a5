"""NN kernel variants (TACO_GEMM2_VARIANT = <BK>x<stages>) on a few plain shapes, steady state (GPU box)."""
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


for (M, N, K) in [(4096, 4096, 4096), (11520, 1024, 256), (11520, 256, 1024), (11520, 1024, 1024), (6400, 2048, 128)]:
    A = torch.randn(M, K, device='cuda'); W = torch.randn(K, N, device='cuda') * 0.05; C = torch.empty(M, N, device='cuda')
    gf = 2.0 * M * N * K / 1e9
    out = []
    for v in sys.argv[1:] or ['32x2', '32x3', '16x3', '16x4']:
        os.environ['TACO_GEMM2_VARIANT'] = v
        us = steady(lambda: lib.conv_gemm(A, W, C, M, N, K, taps=1, T=M, pad_l=0, act=0))
        out.append('%s %.1f us %.1f TF' % (v, us, gf / us * 1e3))
    ub = steady(lambda: torch.mm(A, W))
    print('M=%d N=%d K=%d: %s | BLAS %.1f us %.1f TF' % (M, N, K, ' | '.join(out), ub, gf / ub * 1e3), flush=True)
