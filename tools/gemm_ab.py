"""A/B of the NN conv-GEMM kernels at the model's real shapes (GPU box): old (gemm.hip) vs gemm2 variants, interleaved
rounds in ONE process (guide rule 24).  usage: python tools/gemm_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_amd import lib

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3  # us

NN = [  # name, M, T, N, K, taps
    ('enc proj1 1tap', 6400, 200, 128, 2048, 1),
    ('enc proj1 fwd', 6400, 200, 128, 2048, 3),
    ('post proj1 fwd', 11520, 360, 256, 1024, 3),
    ('enc dpool (bwd)', 6400, 200, 2048, 128, 3),
    ('post dpool (bwd)', 11520, 360, 1024, 256, 3),
    ('post dense bwd', 11520, 11520, 256, 1028, 1),
    ('bank k=16 alone', 6400, 200, 128, 128, 16),
    ('post xproj', 11520, 11520, 768, 128, 1),
    ('enc prenet', 6400, 6400, 256, 256, 1),
    ('square 4096', 4096, 4096, 4096, 4096, 1),
]
VARS = [('old', {'TACO_GEMM2_MIN_TILES': '0'}), ('32x2', None), ('32x3', None), ('16x3', None), ('ksplit', None)]
for name, M, T, N, K, taps in NN:
    A = torch.randn(M, K, device='cuda'); W = torch.randn(taps, K, N, device='cuda') * 0.05; C = torch.empty(M, N, device='cuda')
    slabs = torch.empty(4 * max(6400 * 128, 11520 * 256), device='cuda')
    gf = 2.0 * M * N * K * taps / 1e9
    best = {v: 1e30 for v, _ in VARS}
    for rnd in range(3):
        for v, env in VARS:
            if env: os.environ.update(env)
            else:
                os.environ['TACO_GEMM2_MIN_TILES'] = '1'; os.environ['TACO_GEMM2_VARIANT'] = v
            if v == 'ksplit':
                os.environ['TACO_GEMM2_VARIANT'] = '32x2'
                us = timeit(lambda: lib.conv_gemm_ksplit(A, W, C, M, N, K, slabs, taps=taps, T=T, pad_l=(taps - 1) // 2, act=1))
            else:
                us = timeit(lambda: lib.conv_gemm(A, W, C, M, N, K, taps=taps, T=T, pad_l=(taps - 1) // 2, act=1))
            best[v] = min(best[v], us)
    print('NN %-18s M=%5d N=%4d K=%4d taps=%2d  ' % (name, M, N, K, taps) + '  '.join('%s %7.1f us %5.1f TF' % (v, best[v], gf / (best[v] * 1e-6) / 1e3) for v, _ in VARS), flush=True)

TN = [('enc proj1 dW', 6400, 200, 128, 2048, 3), ('post proj1 dW', 11520, 360, 256, 1024, 3), ('dec gru gates dW', 5760, 180, 512, 256, 1),
      ('bank k=16 dW', 6400, 200, 128, 128, 16), ('highway dW', 6400, 6400, 128, 128, 1), ('post dense dW(1024)', 11520, 11520, 1024, 256, 1)]
for name, M, T, N, K, taps in TN:
    A = torch.randn(M, K, device='cuda'); Y = torch.randn(M, N, device='cuda'); dW = torch.zeros(taps, K, N, device='cuda')
    gf = 2.0 * M * N * K * taps / 1e9
    best = {}
    for rnd in range(3):
        for v, on in (('old', '0'), ('tn2', '1')):
            os.environ['TACO_GEMM2_MIN_TILES'] = '160'
            os.environ['TACO_TN2'] = on
            us = timeit(lambda: lib.gemm_tn(A, Y, dW, M, N, K, taps=taps, T=T, pad_l=(taps - 1) // 2, accumulate=True))
            best[v] = min(best.get(v, 1e30), us)
    print('TN %-18s M=%5d N=%4d K=%4d taps=%2d  ' % (name, M, N, K, taps) + '  '.join('%s %7.1f us %5.1f TF' % (v, best[v], gf / (best[v] * 1e-6) / 1e3) for v in best), flush=True)
