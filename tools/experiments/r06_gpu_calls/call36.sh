#!/bin/bash
# round 6, GPU call 36: bi-GRU recurrences with their scalar LDS operands read ahead of the products (forward: with the h vector; backward: one step ahead)
mkdir -p gpurun_out
{
echo "== parity (bi-GRU op tests + model tests that exercise both kernels)"
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "gru" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 1500 python -m pytest tests/test_gpu_model.py -x -q -k "golden or full_size or medium or stages or speaker" 2>&1 | grep -v amdgpu.ids | tail -2
for rep in 1 2 3; do
  for n in hip prevgru; do TACO_LIB=$PWD/tacotron_amd/libtaco_$n.so timeout 300 python tools/gru_quick.py 2>&1 | grep -v amdgpu.ids | tail -1; done
done
} > gpurun_out/r06_call36.log 2>&1
cat gpurun_out/r06_call36.log | tail -12
