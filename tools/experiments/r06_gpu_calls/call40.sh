#!/bin/bash
# round 6, GPU call 40: weight gradients with the taps merged into K where K is no multiple of the 64-row tile (post-net conv bank, K = 80)
mkdir -p gpurun_out
{
echo "== parity (op tests of the weight-gradient kernel + model tests)"
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "gemm_tn" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 1700 python -m pytest tests/test_gpu_model.py -x -q -k "golden or full_size or medium or deterministic or speaker or trajectory" 2>&1 | grep -v amdgpu.ids | tail -2
for rep in 1 2; do
  for n in hip prevtn; do
    echo "== $n"; TACO_LIB=$PWD/tacotron_amd/libtaco_$n.so timeout 300 python tools/family_trace.py 2>&1 | grep -v amdgpu.ids | grep "^step\|#27\|#29\|#31\|#45\|^sum"
  done
done
bash tools/ab_run.sh hip prevtn
} > gpurun_out/r06_call40.log 2>&1
tail -40 gpurun_out/r06_call40.log
