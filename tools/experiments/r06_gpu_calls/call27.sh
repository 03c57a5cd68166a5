#!/bin/bash
# round 6, GPU call 27: + BPTT round DQ gather shadow (d out / dx rows of round OUT), dropout scales in SGPRs
mkdir -p gpurun_out
{
echo "== parity on the new build (decoder-related model tests)"
timeout 1700 python -m pytest tests/test_gpu_model.py -x -q 2>&1 | grep -v amdgpu.ids | tail -3
bash tools/ab_run.sh hip base
} > gpurun_out/r06_call27.log 2>&1
tail -12 gpurun_out/r06_call27.log
