#!/bin/bash
# round 5, GPU call 1: bf16x3 form of gemm2 -- op parity, per-launch family trace with both forms, model parity, bench
mkdir -p gpurun_out
{
echo "== ops"; timeout 600 python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -15
echo "== family trace BF16X=0"; TACO_GEMM2_BF16X=0 timeout 300 python tools/family_trace.py 2>&1 | grep -v amdgpu.ids
echo "== family trace BF16X=1"; TACO_GEMM2_BF16X=1 timeout 300 python tools/family_trace.py 2>&1 | grep -v amdgpu.ids
echo "== family trace BF16X=1 16x5"; TACO_GEMM2_BF16X=1 TACO_GEMM2_VARIANT=16x5 timeout 300 python tools/family_trace.py 2>&1 | grep -v amdgpu.ids | grep "step\|sum\|nn n=16\|ksplit\|bank-gather\|pool2"
echo "== model tests"; timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_sizes.py -x -q 2>&1 | tail -15
echo "== bench"; timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1
} > gpurun_out/r05_call1.log 2>&1
tail -c 6000 gpurun_out/r05_call1.log
