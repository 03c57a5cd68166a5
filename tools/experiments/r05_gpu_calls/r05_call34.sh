#!/bin/bash
mkdir -p gpurun_out
{
echo "== A/B E tail"; timeout 400 bash tools/ab_run.sh hip nopark notail noboth
} > gpurun_out/r05_call34.log 2>&1
cat gpurun_out/r05_call34.log | tail -70
