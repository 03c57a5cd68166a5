#!/bin/bash
# round 6, GPU call 41: LLVM machine-scheduling strategies for gemm2.hip (NN kernel) and gemm.hip (weight gradients, small NN): family-trace sums
mkdir -p gpurun_out
{
for rep in 1 2; do
  for n in hip g2maxilp g2iterativemaxocc g2iterativeilp gmmaxilp gmiterativemaxocc; do
    echo -n "$n: "; TACO_LIB=$PWD/tacotron_amd/libtaco_$n.so timeout 300 python tools/family_trace.py 2>&1 | grep -v amdgpu.ids | grep "^step\|^sum" | tr '\n' ' '; echo
  done
done
} > gpurun_out/r06_call41.log 2>&1
cat gpurun_out/r06_call41.log
