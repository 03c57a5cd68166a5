#!/bin/bash
mkdir -p gpurun_out
{
for v in 1 0 1 0; do echo "== family trace PROJ2_KSPLIT=$v"; TACO_PROJ2_KSPLIT=$v timeout 300 python tools/family_trace.py 2>&1 | grep -E "^step|^sum|N=80 K=256|N=128 K=128 taps=3..3 res|ksplit S=. M=.* N=(80|128) K=(256|128) "; done
echo "== model tests"; timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_sizes.py -x -q 2>&1 | tail -6
} > gpurun_out/r05_call10.log 2>&1
cat gpurun_out/r05_call10.log | tail -40
