#!/bin/bash
# round 6, GPU call 6: why does bench.py's VCTK leg read 8.5 ms when the same model steps in 7.4 ms by itself?
mkdir -p gpurun_out
{
cat > /tmp/vctk_quick.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from tacotron_amd import lib
from tacotron_amd.config import Config
from tacotron_amd.data import synthetic_batch
from tacotron_amd.model import Tacotron
def run(S, Td=180, steps=10):
    c = Config(); c.r, c.vocab_size, c.num_speakers = 2, 60, S
    m = Tacotron(c, synthetic_batch(32, 200, Td, 2, 60, num_speakers=S), train=True, seed=0)
    for _ in range(3): m.step()
    torch.cuda.synchronize(); m.check()
    t0 = time.perf_counter()
    for _ in range(steps): m.step()
    torch.cuda.synchronize(); print('S=%d Td=%d: %.3f ms/step' % (S, Td, (time.perf_counter() - t0) / steps * 1e3), flush=True)
    del m; torch.cuda.empty_cache()
for spec in sys.argv[1:]:
    S, Td = spec.split(':'); run(int(S), int(Td))
PY
echo "-- alone"; python /tmp/vctk_quick.py 109:180 2>&1 | grep "^S="
echo "-- after S1"; python /tmp/vctk_quick.py 1:180 109:180 2>&1 | grep "^S="
echo "-- after S1 and S2"; python /tmp/vctk_quick.py 1:180 1:500 109:180 1:180 2>&1 | grep "^S="
echo "-- bench --speakers 109"; python bench.py --speakers 109 --no-extras --no-inference --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['kernels_ms'])"
} > gpurun_out/r06_call6.log 2>&1
cat gpurun_out/r06_call6.log
