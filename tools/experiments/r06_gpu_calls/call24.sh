#!/bin/bash
# round 6, GPU call 24: weight-gradient kernel with a TWO-stage register prefetch (product build) vs the one-stage form (libtaco_tnpf1.so)
mkdir -p gpurun_out
{
echo "== parity (op tests + model tests that exercise the weight gradients)"
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q -k "tn or weight_grad or gemm" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 1500 python -m pytest tests/test_gpu_model.py -x -q -k "golden or full_size or medium or deterministic" 2>&1 | grep -v amdgpu.ids | tail -2
for rep in 1 2; do
  for n in hip tnpf1; do
    echo "== $n"; TACO_LIB=$PWD/tacotron_amd/libtaco_$n.so timeout 300 python tools/family_trace.py 2>&1 | grep -v amdgpu.ids | grep "^step\| tn\|^sum"
  done
done
bash tools/ab_run.sh hip tnpf1
} > gpurun_out/r06_call24.log 2>&1
tail -60 gpurun_out/r06_call24.log
