#!/bin/bash
# round 6, GPU call 9: backward-prep work (transposes, backward weight images, transposed composites) beside the post-net bi-GRU instead of beside the encoder
mkdir -p gpurun_out
{
echo "== tests"; timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_sizes.py -x -q 2>&1 | grep -v amdgpu.ids | tail -3
for rep in 1 2 3; do
  echo -n "late (default): "; timeout 300 python tools/dec_quick.py --time-only 2>&1 | grep "^S1"
  echo -n "early         : "; TACO_BWD_PREP_EARLY=1 timeout 300 python tools/dec_quick.py --time-only 2>&1 | grep "^S1"
done
} > gpurun_out/r06_call9.log 2>&1
cat gpurun_out/r06_call9.log
