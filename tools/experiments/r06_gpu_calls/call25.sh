#!/bin/bash
# round 6, GPU call 25: decoder rounds with (1) the epilogue's LDS operands read in front of the mat-vec, (2) poll units kept raw until
# the check, (3) the shadow mat-vec pinned in front of the poll loop (product build) vs the previous build (libtaco_base.so), same box
mkdir -p gpurun_out
{
echo "== parity on the new build"
timeout 1700 python -m pytest tests/test_gpu_model.py -x -q 2>&1 | grep -v amdgpu.ids | tail -3
bash tools/ab_run.sh hip base
python tools/dec3_trace.py 32 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r06_call25.log 2>&1
tail -50 gpurun_out/r06_call25.log
