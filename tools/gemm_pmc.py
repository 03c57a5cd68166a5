"""Launches the NN / TN GEMM kernels at representative model shapes in a FIXED order so that a rocprofv3 --pmc pass can be read
per shape (tools/gemm_pmc_read.py assigns the gemm-named dispatches to shapes by launch order).
usage: [rocprofv3 --pmc ... --] python tools/gemm_pmc.py <order.json>"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_amd import lib

REPS = 4
NN = [  # name, M, T, N, K, taps
    ('square 4096', 4096, 4096, 4096, 4096, 1),
    ('square 2048', 2048, 2048, 2048, 2048, 1),
    ('post dpool M11520 N1024 K256 t3', 11520, 360, 1024, 256, 3),
    ('enc dpool M6400 N2048 K128 t3', 6400, 200, 2048, 128, 3),
    ('post dense fwd N1028 K256', 11520, 11520, 1028, 256, 1),
    ('post dense bwd N256 K1028', 11520, 11520, 256, 1028, 1),
    ('post xproj N768 K128', 11520, 11520, 768, 128, 1),
    ('enc xproj N768 K128', 6400, 6400, 768, 128, 1),
    ('enc proj2 N128 K256 t3', 6400, 200, 128, 256, 3),
    ('post proj2 N80 K256 t3', 11520, 360, 80, 256, 3),
    ('gh M6400 N128 K768', 6400, 6400, 128, 768, 1),
    ('gh M11520 N128 K768', 11520, 11520, 128, 768, 1),
]
TN = [
    ('tn post proj1 dW', 11520, 360, 256, 1024, 3),
    ('tn enc proj1 dW', 6400, 200, 128, 2048, 3),
    ('tn post dense dW(1024)', 11520, 11520, 1024, 256, 1),
    ('tn bank k=16 dW', 6400, 200, 128, 128, 16),
]
order = []
ev = []
for name, M, T, N, K, taps in NN:
    A = torch.randn(M, K, device='cuda'); W = torch.randn(taps, K, N, device='cuda') * 0.05; C = torch.empty(M, N, device='cuda')
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lib.conv_gemm(A, W, C, M, N, K, taps=taps, T=T, pad_l=(taps - 1) // 2, act=1)
    s.record()
    for _ in range(REPS - 1):
        lib.conv_gemm(A, W, C, M, N, K, taps=taps, T=T, pad_l=(taps - 1) // 2, act=1)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / (REPS - 1) * 1e3
    order.append({'name': name, 'n': REPS, 'gflop': 2.0 * M * N * K * taps / 1e9, 'us': us})
for name, M, T, N, K, taps in TN:
    A = torch.randn(M, K, device='cuda'); Y = torch.randn(M, N, device='cuda'); dW = torch.zeros(taps, K, N, device='cuda')
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lib.gemm_tn(A, Y, dW, M, N, K, taps=taps, T=T, pad_l=(taps - 1) // 2, accumulate=True)
    s.record()
    for _ in range(REPS - 1):
        lib.gemm_tn(A, Y, dW, M, N, K, taps=taps, T=T, pad_l=(taps - 1) // 2, accumulate=True)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / (REPS - 1) * 1e3
    order.append({'name': name, 'n': REPS, 'gflop': 2.0 * M * N * K * taps / 1e9, 'us': us})
json.dump(order, open(sys.argv[1], 'w'))
for o in order:
    print('%-36s %8.1f us %6.1f TF' % (o['name'], o['us'], o['gflop'] / o['us'] * 1e-3 if o['us'] else 0))
