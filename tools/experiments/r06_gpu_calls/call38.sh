#!/bin/bash
# round 6, GPU call 38: forward: the sampling flags as ONE load per lane (row = lane % R) read back with v_readlane, instead of R uniform loads per lane
mkdir -p gpurun_out
{
echo "== parity on the new build (model tests + decoder geometries incl. r = 5, inference)"
timeout 1700 python -m pytest tests/test_gpu_model.py -x -q 2>&1 | grep -v amdgpu.ids | tail -3
bash tools/ab_run.sh hip prev
} > gpurun_out/r06_call38.log 2>&1
cat gpurun_out/r06_call38.log | tail -12
