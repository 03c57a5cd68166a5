#!/bin/bash
# round 6, GPU call 39: long soak of the final build (20,000 steps at S1) + 300 steps of the train driver
mkdir -p gpurun_out
{
echo "== 20,000 train steps on one batch, final build"
timeout 600 python tools/soak.py 20000 2>&1 | grep -v amdgpu.ids
echo "== python -m tacotron_amd.train (synthetic corpus in HBM), 300 steps"
timeout 300 python -m tacotron_amd.train -t nancy --steps 300 2>&1 | grep -v amdgpu.ids | tail -6
} > gpurun_out/r06_soak_long_final.txt 2>&1
tail -20 gpurun_out/r06_soak_long_final.txt
