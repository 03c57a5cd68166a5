#!/bin/bash
mkdir -p gpurun_out
{
echo "== ops"; timeout 900 python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -6
echo "== family trace, slabs on"; timeout 300 python tools/family_trace.py 2>&1 | grep -E "^step|^sum| tn"
echo "== family trace, slabs off"; TACO_TN_SLABS=0 timeout 300 python tools/family_trace.py 2>&1 | grep -E "^step|^sum| tn"
echo "== rest of gpu tests"; timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_ops.py 2>&1 | tail -8
echo "== traj"; timeout 300 python -m pytest tests/test_gpu_model.py -q -s -k trajectory 2>&1 | grep "20 steps"
echo "== fabric"; python -c "
from tacotron_amd import lib
print(lib.fabric_probe())" 2>&1 | grep -v amdgpu
} > gpurun_out/r05_call8.log 2>&1
cat gpurun_out/r05_call8.log | tail -60
