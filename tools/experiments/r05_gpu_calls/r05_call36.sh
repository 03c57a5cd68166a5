#!/bin/bash
mkdir -p gpurun_out
{
echo "== A/B E tail, take 2"; timeout 400 bash tools/ab_run.sh hip parkopq notail opqnotail
} > gpurun_out/r05_call36.log 2>&1
cat gpurun_out/r05_call36.log | tail -70
