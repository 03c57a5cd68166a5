"""Debug aid (GPU box): run the medium test shape forward+backward and dump gstash / stash / grads for an A/B between two
library builds.  usage: TACO_LIB=<path> python tools/dump_bwd.py <tag> [nosample]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from tacotron_amd import lib
import test_gpu_model as T
tag = sys.argv[1]
r, V, B, Tt, Td = 2, 40, 4, 37, 12
p = T.on.init_params(V, r, seed=4, perturb=0.2)
inp, masks = T.small_case(r=r, V=V, B=B, Tt=Tt, Td=Td, seed=8)
if len(sys.argv) > 2: masks.pop('sample', None)
R = T.Runner(lib, B, Tt, Td, r, V)
R.set(p, inp, masks); R.forward(); R.backward()
out = {'grads': R.grads.cpu().numpy(), 's2s': R.s2s.cpu().numpy(), 'al': R.al.cpu().numpy()}
for n in ('bwd.gstash', 'dec.stash', 'bwd.dvalues', 'bwd.dkeys', 'bwd.comp.g'):
    out[n] = R.wsget(n)
np.savez(os.path.join('/tmp', 'dump_%s.npz' % tag), **out)
out.pop('grads'); np.savez(os.path.join('gpurun_out', 'dump_%s.npz' % tag), **out)
print(tag, 'saved', {k: v.shape for k, v in out.items()})
