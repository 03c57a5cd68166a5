"""Error of the bf16x3 NN kernel (gemm2.hip) vs the fp32 MFMA form as a function of the CHAIN length (k-products accumulated in one
accumulator), for adversarial (same sign, all mantissa bits set), realistic (post-ReLU activations x glorot weights) and heavy-tailed
operands.  usage: python tools/bf16x3_chain_probe.py   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['TACO_GEMM2_MIN_TILES'] = '1'
os.environ['TACO_GEMM2_VARIANT'] = '16x4'
import numpy as np, torch
from tacotron_amd import lib


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to('cuda', torch.float32).contiguous()


def ones_mantissa(rng, shape, spread=2):
    e = rng.integers(-spread, spread + 1, shape)
    return (np.float32(2.0 - 2.0 ** -23) * np.exp2(e).astype(np.float32)).astype(np.float32)


rng = np.random.default_rng(23)
M, N = 512, 256
print('%-28s %6s  %10s %10s %7s' % ('operands', 'K', 'fp32 MFMA', 'bf16x3', 'ratio'))
for kind in ('ones-mantissa-same-sign', 'uniform-positive', 'relu-x-glorot', 'gaussian', 'heavy-tailed'):
    for K in (256, 512, 1024, 2048, 4096, 6144):
        if kind == 'ones-mantissa-same-sign':
            A, W = ones_mantissa(rng, (M, K)), ones_mantissa(rng, (1, K, N))
        elif kind == 'uniform-positive':
            A, W = rng.uniform(0.5, 1.5, (M, K)).astype(np.float32), rng.uniform(0.5, 1.5, (1, K, N)).astype(np.float32)
        elif kind == 'relu-x-glorot':
            A = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32)
            W = rng.uniform(-1, 1, (1, K, N)).astype(np.float32) * np.float32(np.sqrt(6.0 / (K + N)))
        elif kind == 'gaussian':
            A, W = rng.standard_normal((M, K)).astype(np.float32), (rng.standard_normal((1, K, N)) / np.sqrt(K)).astype(np.float32)
        else:
            A = (rng.standard_normal((M, K)) * np.exp(rng.standard_normal((M, K)) * 3)).astype(np.float32)
            W = (rng.standard_normal((1, K, N)) * np.exp(rng.standard_normal((1, K, N)) * 3) / np.sqrt(K)).astype(np.float32)
        ref = A.astype(np.float64) @ W[0].astype(np.float64)
        err = {}
        for bx in ('0', '1'):
            os.environ['TACO_GEMM2_BF16X'] = bx
            C = torch.full((M, N), float('nan'), device='cuda')
            lib.conv_gemm(dev(A), dev(W), C, M, N, K, taps=1, T=M, pad_l=0, act=0)
            err[bx] = np.linalg.norm(C.cpu().numpy() - ref) / np.linalg.norm(ref)
        print('%-28s %6d  %10.2e %10.2e %7.1f' % (kind, K, err['0'], err['1'], err['1'] / err['0']))
