#!/bin/bash
# XCD-aware tile order of the small-tile NN kernel: parity (all op tests), time (family trace, alternating)
mkdir -p gpurun_out
{
echo "== ops tests"; timeout 600 python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -3
for x in 1 0 1 0 1 0; do echo "== family trace TACO_NN_XCD=$x"; TACO_NN_XCD=$x timeout 300 python tools/family_trace.py 2>&1 | grep -E "^step|^sum| nn n=[0-9]+ M=(6400|11520|5760) N=(128|256|80|160) K"; done
} > gpurun_out/r05_call43.log 2>&1
grep -E "^==|^step|^sum|passed|failed" gpurun_out/r05_call43.log
