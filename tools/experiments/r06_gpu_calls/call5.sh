#!/bin/bash
# round 6, GPU call 5: multi-speaker encoder with the adapters inside the highway stack kernels: parity (fixture, 109 speakers at S1), VCTK step A/B
mkdir -p gpurun_out
{
echo "== speaker tests"; timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_sizes.py -x -q -s -k "speaker or vctk or config4" 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|Error|error|assert|flip|rel_l2=.*(e-0[1-4])" | tail -20
echo "== single-speaker regression (highway kernels changed)"; timeout 1200 python -m pytest tests/test_gpu_model.py -x -q -k "medium or golden or full_size" 2>&1 | grep -v amdgpu.ids | tail -3
cat > /tmp/vctk_quick.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from tacotron_amd import lib
from tacotron_amd.config import Config
from tacotron_amd.data import synthetic_batch
from tacotron_amd.model import Tacotron
S = int(sys.argv[1])
c = Config(); c.r, c.vocab_size, c.num_speakers = 2, 60, S
m = Tacotron(c, synthetic_batch(32, 200, 180, 2, 60, num_speakers=S), train=True, seed=0)
for _ in range(3): m.step()
torch.cuda.synchronize(); m.check()
t0 = time.perf_counter()
for _ in range(10): m.step()
torch.cuda.synchronize(); print('S=%d: %.3f ms/step loss %.1f' % (S, (time.perf_counter() - t0) / 10 * 1e3, float(m.loss)))
PY
for rep in 1 2 3; do
  python /tmp/vctk_quick.py 1 2>&1 | grep "^S="
  python /tmp/vctk_quick.py 109 2>&1 | grep "^S="
  echo -n "unfused "; TACO_SPK_UNFUSED=1 python /tmp/vctk_quick.py 109 2>&1 | grep "^S="
done
} > gpurun_out/r06_call5.log 2>&1
tail -c 4000 gpurun_out/r06_call5.log
