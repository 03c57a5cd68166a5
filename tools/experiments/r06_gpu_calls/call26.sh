#!/bin/bash
# round 6, GPU call 26: + C2's x read in the poll shadow, BPTT rider preloads, rounds G1 / G2 with the input half of the candidate mat-vec in their poll shadow
mkdir -p gpurun_out
{
echo "== parity on the new build (decoder-related model tests)"
timeout 1700 python -m pytest tests/test_gpu_model.py -x -q 2>&1 | grep -v amdgpu.ids | tail -3
bash tools/ab_run.sh hip base
} > gpurun_out/r06_call26.log 2>&1
tail -12 gpurun_out/r06_call26.log
