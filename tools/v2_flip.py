import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import taco_numpy as on
from tests.util import small_case
from tests.test_gpu_model import Runner
from tacotron_amd import lib
L = ctypes.CDLL(lib.LIB_PATH)
r, V, B, Tt, Td = 2, 40, 4, 37, 12
p = on.init_params(V, r, seed=4, perturb=0.2)
inp, masks = small_case(r=r, V=V, B=B, Tt=Tt, Td=Td, seed=8)
os.environ['TACO_GEMM2_MIN_TILES'] = '1'
names = ['enc.th0', 'enc.th1', 'enc.th2', 'enc.th3', 'enc.h1', 'enc.h4', 'enc.p1', 'enc.p2', 'enc.bank', 'enc.pool', 'enc.pj1pre', 'enc.pj1', 'enc.pj2pre', 'enc.res']
def run(lo, hi):
    L.taco_debug_gemm2_window(lo, hi)
    R = Runner(lib, B, Tt, Td, r, V)
    R.set(p, inp, masks); R.forward()
    out = {n: R.wsget(n).copy() for n in names}
    R.backward()
    L.taco_debug_gemm2_window(0, 1 << 30)
    return out, R.pb.to_dict(R.grads)['embedding']
a, ga = run(0, 0)
b, gb = run(2, 7)
print('embedding grad diff %.2e' % (np.linalg.norm(ga - gb) / np.linalg.norm(ga)))
for n in names:
    x, y = a[n], b[n]
    flips = np.argwhere((x == 0) != (y == 0))
    print('%-10s max|d| %.2e  zero/non-zero flips: %d' % (n, np.abs(x - y).max(), len(flips)), [(tuple(int(v) for v in f), float(x[tuple(f)]), float(y[tuple(f)])) for f in flips[:6]])
# max-pool near-ties: where the arg-max side of the pooled BN output differs
bank_a, bank_b = a['enc.bank'], b['enc.bank']
