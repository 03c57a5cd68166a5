#!/bin/bash
mkdir -p gpurun_out
{
echo "== decoder parity (model tests)"; timeout 900 python -m pytest tests/test_gpu_model.py -x -q 2>&1 | tail -5
echo "== A/B prein hoisted (hip) / parked loads hot / none"; timeout 400 bash tools/ab_run.sh hip parkhot nopark
} > gpurun_out/r05_call35.log 2>&1
cat gpurun_out/r05_call35.log | tail -70
