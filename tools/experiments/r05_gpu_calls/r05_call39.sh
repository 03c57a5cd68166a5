#!/bin/bash
mkdir -p gpurun_out
{
echo "== scalar wave index"; timeout 600 bash tools/ab_run.sh hip swave
} > gpurun_out/r05_call39.log 2>&1
cat gpurun_out/r05_call39.log | tail -70
