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
def run(lo, hi):
    L.taco_debug_gemm2_window(lo, hi)
    R = Runner(lib, B, Tt, Td, r, V)
    R.set(p, inp, masks); R.forward(); R.backward()
    n = L.taco_debug_gemm2_window(0, 1 << 30)
    return R.pb.to_dict(R.grads)['embedding'], n
os.environ['TACO_GEMM2_TRACE'] = '1'
base, n = run(0, 0)
del os.environ['TACO_GEMM2_TRACE']
print('eligible launches fwd+bwd %d' % n)
def diff(lo, hi):
    g, _ = run(lo, hi)
    return np.linalg.norm(g - base) / np.linalg.norm(base)
print('all: %.2e' % diff(0, n))
for a, b in ((2, 3), (6, 7), (2, 7)):
    print('window [%d,%d): %.2e' % (a, b, diff(a, b)))
g1, _ = run(2, 7)
print('nonzero diff rows of embedding grad (V=%d rows):' % V, np.nonzero(np.abs(g1 - base).max(1) > 1e-5 * np.abs(base).max())[0])
