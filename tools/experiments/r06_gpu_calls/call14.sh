#!/bin/bash
# round 6, GPU call 14: bf16x3 with a SECOND accumulator for the five low-order plane products (lab build -DTACO_BF16X_ACC2, gemm2.hip only):
# accuracy vs chain length with the chain bound lifted, kernel time, family trace
mkdir -p gpurun_out
{
echo "== chain probe, bound lifted: product build (one accumulator)"; TACO_BF16X_MAX_CHAIN=1000000000 timeout 200 python tools/bf16x3_chain_probe.py 2>&1 | grep -v amdgpu.ids | grep -E "operands|ones-mantissa|uniform-positive|relu"
echo "== chain probe, bound lifted: two accumulators"; TACO_LIB=$PWD/tacotron_amd/libtaco_acc2.so TACO_BF16X_MAX_CHAIN=1000000000 timeout 200 python tools/bf16x3_chain_probe.py 2>&1 | grep -v amdgpu.ids
for rep in 1 2; do
echo "== lab timing: product"; timeout 200 python tools/gemm_lab6.py 2>&1 | grep -v amdgpu.ids
echo "== lab timing: two accumulators"; TACO_LIB=$PWD/tacotron_amd/libtaco_acc2.so timeout 200 python tools/gemm_lab6.py 2>&1 | grep -v amdgpu.ids
echo "== family: product"; timeout 300 python tools/family_trace.py 2>&1 | grep -E "^step|^sum"
echo "== family: two accumulators"; TACO_LIB=$PWD/tacotron_amd/libtaco_acc2.so timeout 300 python tools/family_trace.py 2>&1 | grep -E "^step|^sum"
done
} > gpurun_out/r06_acc2.txt 2>&1
cat gpurun_out/r06_acc2.txt
