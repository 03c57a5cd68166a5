"""Socket power / shader clock (rocm-smi) sampled while a load runs (GPU box):
   idle, the library's fp32 4096^3 GEMM in a loop, torch.mm fp32 (hipBLASLt/rocBLAS) in a loop, a bf16 torch.mm loop, the train step.
usage: python tools/power_probe.py > gpurun_out/r04_power.txt"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_amd import lib
from tacotron_amd.config import Config
from tacotron_amd.data import synthetic_batch
from tacotron_amd.model import Tacotron


def smi():
    out = []
    for cmd in (['rocm-smi', '--showpower', '--showclocks', '--showmaxpower', '--showperflevel'],):
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=20)
            out += [l.strip() for l in r.stdout.splitlines() if any(k in l for k in ('Power', 'sclk', 'mclk', 'fclk', 'Performance'))]
        except Exception as e:   # noqa
            out.append('rocm-smi failed: %r' % (e,))
    return out


def under(name, fn, seconds=4.0):
    stop = [False]
    count = [0]

    def loop():
        while not stop[0]:
            fn()
            count[0] += 1
        torch.cuda.synchronize()
    th = threading.Thread(target=loop)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th.start()
    time.sleep(seconds / 2)
    s1 = smi()
    time.sleep(seconds / 2)
    stop[0] = True
    th.join()
    dt = time.perf_counter() - t0
    print('== %s: %d iterations in %.2f s (%.3f ms each)' % (name, count[0], dt, dt / max(count[0], 1) * 1e3))
    for l in s1:
        print('   ', l)
    sys.stdout.flush()
    return dt / max(count[0], 1)


def main():
    print('== idle')
    for l in smi():
        print('   ', l)
    n = 4096
    A = torch.randn(n, n, device='cuda'); W = torch.randn(1, n, n, device='cuda') * 0.05; C = torch.empty(n, n, device='cuda')

    def ours():
        for _ in range(20):
            lib.conv_gemm(A, W, C, n, n, n, taps=1, T=n, pad_l=0, act=0)
        torch.cuda.synchronize()
    t = under('library NN kernel, fp32 4096^3 x20', ours)
    print('    -> %.1f TFLOP/s' % (20 * 2.0 * n ** 3 / t / 1e12))
    B2 = W[0]

    def blas():
        for _ in range(20):
            torch.mm(A, B2, out=C)
        torch.cuda.synchronize()
    t = under('torch.mm fp32 4096^3 x20 (vendor BLAS)', blas)
    print('    -> %.1f TFLOP/s' % (20 * 2.0 * n ** 3 / t / 1e12))
    Ah, Bh = A.bfloat16(), B2.bfloat16(); Ch = torch.empty(n, n, device='cuda', dtype=torch.bfloat16)

    def blas16():
        for _ in range(100):
            torch.mm(Ah, Bh, out=Ch)
        torch.cuda.synchronize()
    t = under('torch.mm bf16 4096^3 x100 (vendor BLAS)', blas16)
    print('    -> %.1f TFLOP/s' % (100 * 2.0 * n ** 3 / t / 1e12))
    c = Config(); c.r, c.vocab_size = 2, 60
    m = Tacotron(c, synthetic_batch(32, 200, 180, 2, 60), train=True, seed=0)
    for _ in range(3): m.step()

    def step():
        for _ in range(10): m.step()
        torch.cuda.synchronize()
    t = under('train step x10', step)
    print('    -> %.3f ms per step; clock probe %s' % (t / 10 * 1e3, lib.clock_probe() if hasattr(lib, 'clock_probe') else ''))
    m.check()


if __name__ == '__main__':
    main()
