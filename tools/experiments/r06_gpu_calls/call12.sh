#!/bin/bash
# round 6, GPU call 12: round E energies -> softmax from the poll registers, for R <= 2 only (inference B <= 16; the R = 4 kernels have no registers for it)
mkdir -p gpurun_out
{
echo "== parity (fixtures run R = 1; sizes run B = 1 / 2 / 32)"; timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_sizes.py tests/test_gpu_frontend.py -x -q 2>&1 | grep -v amdgpu.ids | tail -3
for rep in 1 2 3; do
  echo -n "regsoftmax (R<=2): "; timeout 200 python tools/infer_ab.py 2>&1 | grep "^B="
  echo -n "round-5 form      : "; TACO_LIB=$PWD/tacotron_amd/libtaco_noesr.so timeout 200 python tools/infer_ab.py 2>&1 | grep "^B="
done
} > gpurun_out/r06_call12.log 2>&1
cat gpurun_out/r06_call12.log
