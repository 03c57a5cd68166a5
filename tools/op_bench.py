"""Times the bi-GRU recurrences at the model's shapes (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_amd import lib
for B, T in ((32, 200), (32, 360)):
    w = {}
    for d in ('fw', 'bw'):
        w[d + '/gates/kernel'] = (torch.rand(256, 256, device='cuda') - 0.5) * 0.3
        w[d + '/gates/bias'] = torch.ones(256, device='cuda')
        w[d + '/candidate/kernel'] = (torch.rand(256, 128, device='cuda') - 0.5) * 0.3
        w[d + '/candidate/bias'] = torch.zeros(128, device='cuda')
    x = torch.randn(B, T, 128, device='cuda'); xg = torch.zeros(B, T, 768, device='cuda')
    out = torch.zeros(B, T, 256, device='cuda'); ruc = torch.zeros(B, T, 768, device='cuda')
    for _ in range(3): lib.bigru_fwd(x, w, xg, out, ruc, B, T)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): lib.bigru_fwd(x, w, xg, out, ruc, B, T)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 10 * 1e3
    print('bigru_fwd (incl. x-proj GEMM) B=%d T=%d: %.1f us  (%.2f us/step)' % (B, T, us, us / T))
