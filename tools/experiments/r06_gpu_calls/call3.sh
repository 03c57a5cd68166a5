#!/bin/bash
# round 6, GPU call 3: pre-split weight images (B-image form of gemm2): op parity (bit-identical to the in-register split), adversarial
# tests with the chain bound, family trace A/B (BSPLIT 0 / 1, ring 3 / 4 stages), S1 step A/B, model parity
mkdir -p gpurun_out
{
echo "== ops: image form + adversarial"
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "weight_image or adversarial or chain_bound or bf16x3" 2>&1 | grep -v amdgpu.ids | tail -8
for cfg in "BSPLIT=0 NS=3" "BSPLIT=1 NS=3" "BSPLIT=1 NS=4" "BSPLIT=0 NS=3" "BSPLIT=1 NS=3" "BSPLIT=1 NS=4"; do
  eval $cfg; echo "== family trace TACO_GEMM2_BSPLIT=$BSPLIT TACO_GEMM2_BI_NS=$NS"
  TACO_GEMM2_BSPLIT=$BSPLIT TACO_GEMM2_BI_NS=$NS timeout 300 python tools/family_trace.py 2>&1 | grep -E "^step|^sum|nn (n=16|n=8|n=1 M=11520 N=10|n=1 M=11520 N=256 K=1028|bank|n=4)|nn-ksplit|pool2|N=128 K=768"
done
echo "== model tests"; timeout 1500 python -m pytest tests/test_gpu_model.py -x -q 2>&1 | grep -v amdgpu.ids | tail -6
} > gpurun_out/r06_call3.log 2>&1
tail -c 9000 gpurun_out/r06_call3.log
