#!/bin/bash
# round 6, GPU call 18: B > 32 on decoder3.hip as consecutive launches of <= 32 rows: parity (B = 40, 50, 70), then timing at B = 48 / 64 vs the decoder.hip fallback
mkdir -p gpurun_out
{
echo "== geometries"; timeout 1500 python -m pytest tests/test_gpu_model.py -x -q -k "cluster_geometries" 2>&1 | grep -v amdgpu.ids | tail -3
cat > /tmp/bq.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tacotron_amd import lib
from tacotron_amd.config import Config
from tacotron_amd.data import synthetic_batch
from tacotron_amd.model import Tacotron
for B in (32, 48, 64):
    c = Config(); c.r, c.vocab_size = 2, 60
    m = Tacotron(c, synthetic_batch(B, 200, 180, 2, 60), train=True, seed=0)
    for _ in range(3): m.step()
    torch.cuda.synchronize(); m.check()
    lib.profile_read(0); lib.profile_read(1); lib.profile_enable(3)
    t0 = time.perf_counter()
    for _ in range(8): m.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
    lib.profile_enable(0); f, b = lib.profile_read(0), lib.profile_read(1); m.check()
    print('B=%d: %.2f ms/step = %.3f M frames/s; decoder fwd %.2f ms bwd %.2f ms; cluster %d' % (B, dt * 1e3, B * 360 / dt / 1e6, np.median(f), np.median(b), lib.last_cluster(0)), flush=True)
    del m; torch.cuda.empty_cache()
PY
echo "== decoder3, chunked"; python /tmp/bq.py 2>&1 | grep "^B="
echo "== decoder.hip fallback (TACO_DEC_V3=0)"; TACO_DEC_V3=0 python /tmp/bq.py 2>&1 | grep "^B="
} > gpurun_out/r06_call18.log 2>&1
cat gpurun_out/r06_call18.log
