#!/bin/bash
# wave-uniform poll loop (product build) vs the per-lane form (libtaco_perlane.so): parity of the decoder tests, then timing A/B
mkdir -p gpurun_out
{
echo "== decoder parity (model tests)"; timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_sizes.py -x -q 2>&1 | tail -5
echo "== A/B"; timeout 400 bash tools/ab_run.sh hip perlane
} > gpurun_out/r05_call32.log 2>&1
cat gpurun_out/r05_call32.log | tail -30
