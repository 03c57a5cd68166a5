#!/bin/bash
# round 6, GPU call 37: full GPU suite + the round's profile set on the final build
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r06_final_gpu_tests.txt
cat gpurun_out/r06_final_gpu_tests.txt
bash tools/profile_round.sh r06 > gpurun_out/r06_profile_round.log 2>&1
tail -20 gpurun_out/r06_profile_round.log
