#!/bin/bash
mkdir -p gpurun_out
{
echo "== trace B=32 train"; timeout 200 python tools/dec3_trace.py 32 2>&1 | grep -v amdgpu
echo "== trace B=1 infer"; timeout 200 python tools/dec3_trace.py 1 infer 2>&1 | grep -v amdgpu
echo "== A/B"; timeout 400 bash tools/ab_run.sh hip nosleep
} > gpurun_out/r05_call33.log 2>&1
cat gpurun_out/r05_call33.log | tail -70
