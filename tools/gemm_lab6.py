"""Round-6 lab for the B-image form of conv_gemm2 (BX = 2): plain shapes, steady state, with the weight image registered.
usage: TACO_LIB=<lab build> python tools/gemm_lab6.py      (GPU box; lab builds: tools/ab_build.sh <name> "-DGEMM2_LAB_..." gemm2)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['TACO_BF16X_MAX_CHAIN'] = str(1 << 30)   # (the lab times the bf16x3 forms at any depth)
os.environ['TACO_GEMM2_MIN_TILES'] = '1'
import torch
from tacotron_amd import lib


def steady(fn, seconds=0.4):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize(); n += 20
    return (time.perf_counter() - t0) / n * 1e6


out = []
for (M, N, K) in [(4096, 4096, 4096), (11520, 1024, 256), (11520, 256, 1024), (6400, 2048, 128), (6400, 128, 2048)]:
    A = torch.randn(M, K, device='cuda'); W = torch.randn(K, N, device='cuda') * 0.05; C = torch.empty(M, N, device='cuda')
    gf = 2.0 * M * N * K / 1e9
    row = []
    for form in ('img3', 'img4', 'reg', 'f32'):
        lib.weight_image(None)
        os.environ['TACO_GEMM2_BF16X'] = '0' if form == 'f32' else '1'
        os.environ['TACO_GEMM2_BI_NS'] = '4' if form == 'img4' else '3'
        img = lib.weight_image(W.view(1, K, N), taps=1, K=K, N=N) if form.startswith('img') else None
        if form == 'img4':
            continue   # (BI_NS is read once per process: run the tool twice with TACO_GEMM2_BI_NS=4 for that column)
        us = steady(lambda: lib.conv_gemm(A, W, C, M, N, K, taps=1, T=M, pad_l=0, act=0))
        row.append('%s %.1f us %.0f TFe' % (form, us, gf / us * 1e3))
        del img
    print('M=%d N=%d K=%d: %s' % (M, N, K, ' | '.join(row)), flush=True)
