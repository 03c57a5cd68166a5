#!/bin/bash
# round 6, GPU call 32: forward stash stores through a buffer descriptor (lane-constant offsets, no exec masking, no 64-bit address arithmetic) + softmax tidy-up
mkdir -p gpurun_out
{
echo "== parity on the new build"
timeout 1700 python -m pytest tests/test_gpu_model.py -x -q 2>&1 | grep -v amdgpu.ids | tail -3
bash tools/ab_run.sh hip prev
} > gpurun_out/r06_call32.log 2>&1
cat gpurun_out/r06_call32.log | tail -12
