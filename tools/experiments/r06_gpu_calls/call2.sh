#!/bin/bash
# round 6, GPU call 2: what v_mfma_f32_32x32x16_bf16 does with products that are small against its accumulator; the rest of call 1's tests
mkdir -p gpurun_out
{
echo "== mfma probe"; timeout 60 ./build/mfma_bf16_probe
echo "== new tests (no -x)"
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -s -k "adversarial or attention_scores or sticky or optional_paths" 2>&1 | grep -v "amdgpu.ids" | grep "rel_l2\|rel-L2\|passed\|failed\|PASS\|FAIL\|Error\|triples\|assert" | head -120
} > gpurun_out/r06_call2.log 2>&1
tail -c 9000 gpurun_out/r06_call2.log
