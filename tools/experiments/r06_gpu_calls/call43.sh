#!/bin/bash
# round 6, GPU call 43: image-form NN kernel with the MFMAs of sub-tiles 2 / 3 interleaved (no instruction between two MFMAs on one accumulator) vs the block form
mkdir -p gpurun_out
{
echo "== parity (NN kernel op tests incl. the bit-identity of the image form)"
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "conv_gemm or bf16x3 or pooled or ksplit" 2>&1 | grep -v amdgpu.ids | tail -2
for rep in 1 2 3; do
  for n in hip nopair; do
    echo -n "$n: "; TACO_LIB=$PWD/tacotron_amd/libtaco_$n.so timeout 300 python tools/family_trace.py 2>&1 | grep -v amdgpu.ids | grep "^step\|# 5 \|#30\|#50\|^sum" | cut -c1-60 | tr '\n' '|'; echo
  done
done
} > gpurun_out/r06_call43.log 2>&1
cat gpurun_out/r06_call43.log
