#!/bin/bash
# round 6, GPU call 17: weight-gradient workgroup target re-swept on the two-accumulator kernels (5 instead of 8 waves per SIMD)
mkdir -p gpurun_out
{
for b in 3072 2048 4096 1536 3072 2048 4096 1536; do echo -n "TACO_TN_BLOCKS=$b: "; TACO_TN_BLOCKS=$b timeout 300 python tools/family_trace.py 2>&1 | grep -E "^step|^sum" | tr '\n' ' '; echo; done
} > gpurun_out/r06_call17.log 2>&1
cat gpurun_out/r06_call17.log
