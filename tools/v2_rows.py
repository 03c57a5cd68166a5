import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tacotron_amd import lib
os.environ['TACO_GEMM2_MIN_TILES'] = '1'
rng = np.random.default_rng(0)
for (M, T, N, K, taps) in [(148, 148, 128, 768, 1), (148, 37, 128, 128, 16), (148, 37, 2048, 128, 3), (148, 148, 256, 128, 1), (512, 512, 128, 256, 1)]:
    A = torch.randn(M, K, device='cuda'); W = torch.randn(taps, K, N, device='cuda') / (K * taps) ** 0.5
    ref = torch.zeros(M, N, device='cuda', dtype=torch.float64)
    A64 = A.double().view(M // T, T, K)
    pad_l = (taps - 1) // 2
    for j in range(taps):
        sh = j - pad_l
        sl = torch.zeros_like(A64)
        if sh >= 0: sl[:, :T - sh] = A64[:, sh:]
        else: sl[:, -sh:] = A64[:, :T + sh]
        ref += (sl.reshape(M, K) @ W[j].double())
    for rep in range(2):
        C = torch.full((M, N), float('nan'), device='cuda')
        lib.conv_gemm(A, W, C, M, N, K, taps=taps, T=T, pad_l=pad_l)
        torch.cuda.synchronize()
        err = (C.double() - ref).abs().max(1).values.cpu().numpy()
        worst = np.argsort(-err)[:6]
        print((M, T, N, K, taps), 'rep', rep, 'row errs[0:6]', ['%.1e' % e for e in err[:6]], 'median %.1e' % np.median(err), 'worst rows', [(int(i), '%.1e' % err[i]) for i in worst])
