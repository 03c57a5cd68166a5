#!/bin/bash
# round 6, GPU call 35: BPTT: prefetched inputs landed in the poll shadow of the previous step last gather (no barrier of their own at the step start)
mkdir -p gpurun_out
{
echo "== parity on the new build (model tests + decoder geometries incl. r = 5, inference)"
timeout 1700 python -m pytest tests/test_gpu_model.py -x -q 2>&1 | grep -v amdgpu.ids | tail -3
bash tools/ab_run.sh hip prev
} > gpurun_out/r06_call35.log 2>&1
cat gpurun_out/r06_call35.log | tail -12
