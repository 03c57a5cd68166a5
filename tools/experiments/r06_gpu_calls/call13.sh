#!/bin/bash
# round 6, GPU call 13: long soak (20,000 train steps on one batch: the weights move far from initialisation -- keys / queries beyond the tanh bound,
# saturating gates) + host driver run
mkdir -p gpurun_out
{
echo "== soak 20000"; timeout 600 python tools/soak.py 20000 2>&1 | grep -v amdgpu.ids
echo "== train driver, 300 steps"; timeout 300 python -m tacotron_amd.train -t nancy --steps 300 2>&1 | grep -v amdgpu.ids | tail -6
} > gpurun_out/r06_soak_long.txt 2>&1
cat gpurun_out/r06_soak_long.txt
