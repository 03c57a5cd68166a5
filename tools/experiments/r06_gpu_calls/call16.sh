#!/bin/bash
# round 6, GPU call 16: two-accumulator bf16x3 as the default: whole GPU suite
mkdir -p gpurun_out
{
echo "== pytest -m gpu"; timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -6
echo "== S1"; for i in 1 2; do timeout 300 python tools/dec_quick.py --time-only 2>&1 | grep "^S1"; done
} > gpurun_out/r06_call16.log 2>&1
cat gpurun_out/r06_call16.log
