#!/bin/bash
# round 6, GPU call 19: whole GPU suite on the final build, then the final profile round
mkdir -p gpurun_out
{
echo "== pytest -m gpu"; timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -4
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
} > gpurun_out/r06_call19.log 2>&1
cat gpurun_out/r06_call19.log
bash tools/profile_round.sh r06 > gpurun_out/r06_profile_round.log 2>&1; tail -3 gpurun_out/r06_profile_round.log
