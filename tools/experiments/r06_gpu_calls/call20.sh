#!/bin/bash
# round 6, GPU call 20: robustness extras -- multi-speaker fixture + full-size config4 under TACO_DETERMINISTIC=1; 2,000-step soak at B = 64 (chunked decoder3)
mkdir -p gpurun_out
{
echo "== speakers, deterministic"; TACO_DETERMINISTIC=1 timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_sizes.py -x -q -k "speaker or config4" 2>&1 | grep -v amdgpu.ids | tail -2
cat > /tmp/soak64.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from tacotron_amd.config import Config
from tacotron_amd.data import synthetic_batch
from tacotron_amd.model import Tacotron
c = Config(); c.r, c.vocab_size = 2, 60
m = Tacotron(c, synthetic_batch(64, 200, 180, 2, 60), train=True, seed=0)
t0 = time.perf_counter(); n = 2000
for i in range(n):
    m.step()
    if i % 500 == 0 or i == n - 1:
        m.check(); print('step %5d  loss %.6g' % (i, float(m.loss)), flush=True)
torch.cuda.synchronize()
print('B=64: %d steps, %.2f ms/step incl. host checks; error words %s; census %s' % (n, (time.perf_counter() - t0) / n * 1e3, m._err.tolist()[:2], m.placement_census()[:2]))
PY
echo "== soak B=64"; timeout 300 python /tmp/soak64.py 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r06_call20.log 2>&1
cat gpurun_out/r06_call20.log
