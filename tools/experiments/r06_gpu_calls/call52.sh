#!/bin/bash
# round 6, GPU call 52: weight-gradient tile / split knobs re-measured on the final build (family traces + step time)
mkdir -p gpurun_out
{
for v in "TACO_NOP=1" "TACO_TN_BM=128" "TACO_TN_BIG_TILES=32" "TACO_TN_BLOCKS=1536" "TACO_TN_BLOCKS=6144"; do
echo "== $v"
env $v timeout 300 python tools/family_trace.py 2>&1 | grep -v amdgpu.ids | grep "^step\|  tn\|^sum" | cut -c1-140
done
} > gpurun_out/r06_call52.log 2>&1
cat gpurun_out/r06_call52.log
