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
    return s.elapsed_time(e) / n * 1e3
slabs = torch.empty(4 * max(6400 * 128, 11520 * 256), device='cuda')
for name, M, T, N, K, taps in [('enc proj1 fwd', 6400, 200, 128, 2048, 3), ('post proj1 fwd', 11520, 360, 256, 1024, 3)]:
    A = torch.randn(M, K, device='cuda'); W = torch.randn(taps, K, N, device='cuda') * 0.05; C = torch.empty(M, N, device='cuda')
    gf = 2.0 * M * N * K * taps / 1e9
    out = []
    for S in (1, 2, 3, 4, 5, 6, 8, 10, 12, 14):
        os.environ['TACO_KSPLIT'] = str(S)
        best = min(timeit(lambda: lib.conv_gemm_ksplit(A, W, C, M, N, K, slabs, taps=taps, T=T, pad_l=1, act=1)) for _ in range(3))
        out.append('S=%d %.1f us %.1f TF' % (S, best, gf / (best * 1e-6) / 1e3))
    print(name, ' | '.join(out), flush=True)
