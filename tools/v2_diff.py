"""Debug: compare the train step with the NN GEMMs on gemm.hip vs gemm2.hip (forced) at the medium test shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import taco_numpy as on
from tests.util import small_case
from tests.test_gpu_model import Runner
from tacotron_amd import lib
r, V, B, Tt, Td = 2, 40, 4, 37, 12
p = on.init_params(V, r, seed=4, perturb=0.2)
inp, masks = small_case(r=r, V=V, B=B, Tt=Tt, Td=Td, seed=8)
res = {}
for mode in ('0', '1'):
    os.environ['TACO_GEMM2_MIN_TILES'] = mode
    R = Runner(lib, B, Tt, Td, r, V)
    R.set(p, inp, masks); R.forward(); R.backward()
    res[mode] = (R.pb.to_dict(R.grads), {k: R.wsget(k) for k in ('bwd.gC', 'bwd.gB', 'bwd.gA', 'bwd.gD', 'bwd.gE', 'bwd.gG')})
g0, w0 = res['0']; g1, w1 = res['1']
for k in g0:
    d = np.linalg.norm(g0[k] - g1[k]) / (np.linalg.norm(g0[k]) + 1e-30)
    if d > 2e-6: print('grad %-45s rel diff %.3e' % (k, d))
for k in w0:
    a, b = w0[k], w1[k]
    d = np.abs(a - b)
    print('ws %-8s shape %s max|d| %.3e at %s  |ref|max %.3e' % (k, a.shape, d.max(), np.unravel_index(d.argmax(), d.shape), np.abs(a).max()))
a, b = w0['bwd.gC'][:B * Tt, :128], w1['bwd.gC'][:B * Tt, :128]
d = np.abs(a - b)
rows = np.argsort(-d.max(1))[:12]
print('dP2 worst rows', [(int(i), int(i) % Tt, float(d[i].max())) for i in rows])
