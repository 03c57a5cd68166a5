#!/bin/bash
# round 6, GPU call 1: the new tests (self-launching bench rehearsal, adversarial bf16x3, tanh precision cases, error word 2), S1 baseline of the box
mkdir -p gpurun_out
{
echo "== new tests"
timeout 1500 python -m pytest tests/test_gpu_dist.py "tests/test_gpu_ops.py" tests/test_gpu_model.py -x -q -s -k "self_launches or two_process or adversarial or bf16x3_products or attention_scores or sticky or optional_paths" 2>&1 | grep -v "amdgpu.ids" | tail -80
echo "== dec_quick"; timeout 300 python tools/dec_quick.py --time-only 2>&1 | grep -v amdgpu.ids
echo "== family trace"; timeout 300 python tools/family_trace.py 2>&1 | grep -v amdgpu.ids
echo "== rehearsal by hand"; timeout 600 python bench.py --gpus 2 --rehearse-shared-device --steps 3 --warmup 1 2>gpurun_out/r06_rehearse.err | tail -1 | cut -c 1-1500
tail -5 gpurun_out/r06_rehearse.err
} > gpurun_out/r06_call1.log 2>&1
tail -c 7000 gpurun_out/r06_call1.log
