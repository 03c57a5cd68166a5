#!/bin/bash
# round 6, GPU call 33: forward next-step inputs re-loaded behind their last use (three groups) instead of together behind round E
mkdir -p gpurun_out
{
echo "== parity on the new build"
timeout 1700 python -m pytest tests/test_gpu_model.py -x -q 2>&1 | grep -v amdgpu.ids | tail -3
bash tools/ab_run.sh hip prev
} > gpurun_out/r06_call33.log 2>&1
cat gpurun_out/r06_call33.log | tail -12
