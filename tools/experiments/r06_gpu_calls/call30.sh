#!/bin/bash
# round 6, GPU call 31: + round OUT stores issued behind the layer-1 check in round E
# 'prev' = the committed build before this change, 'base' = the build of the round-6 profile set
mkdir -p gpurun_out
{
echo "== parity on the new build"
timeout 1700 python -m pytest tests/test_gpu_model.py -x -q 2>&1 | grep -v amdgpu.ids | tail -3
bash tools/ab_run.sh hip prev base
python tools/dec3_trace.py 32 2>&1 | grep -v amdgpu.ids | sed -n '1,13p'
} > gpurun_out/r06_call31.log 2>&1
cat gpurun_out/r06_call31.log | tail -30
