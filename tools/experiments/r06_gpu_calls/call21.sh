#!/bin/bash
# round 6, GPU call 21: decoder weight gradients: the big independent group first (beside the encoder bi-GRU backward), the small dependent launches after
mkdir -p gpurun_out
{
echo "== parity"; timeout 1500 python -m pytest tests/test_gpu_model.py -x -q -k "golden or full_size or medium or trajectory or deterministic" 2>&1 | grep -v amdgpu.ids | tail -2
for rep in 1 2 3; do
  echo -n "tn first (default): "; timeout 300 python tools/dec_quick.py --time-only 2>&1 | grep "^S1"
  echo -n "tn late (round 5)  : "; TACO_DEC_TN_LATE=1 timeout 300 python tools/dec_quick.py --time-only 2>&1 | grep "^S1"
done
} > gpurun_out/r06_call21.log 2>&1
cat gpurun_out/r06_call21.log
