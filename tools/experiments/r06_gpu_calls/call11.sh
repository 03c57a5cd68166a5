#!/bin/bash
# round 6, GPU call 11: the whole GPU suite + smoke on the final build
mkdir -p gpurun_out
{
echo "== pytest -m gpu"; timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -6
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -4
} > gpurun_out/r06_call11.log 2>&1
cat gpurun_out/r06_call11.log
