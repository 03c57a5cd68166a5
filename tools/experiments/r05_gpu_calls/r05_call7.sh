#!/bin/bash
mkdir -p gpurun_out
{
echo "== gpu tests"; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -12
echo "== co-tenant test output"; timeout 300 python -m pytest tests/test_gpu_sizes.py -q -s -k co_tenant 2>&1 | grep "B=1 decoder"
echo "== fabric"; python -c "
from tacotron_amd import lib
print(lib.fabric_probe())" 2>&1 | grep -v amdgpu
echo "== quick S1"; timeout 200 python tools/dec_quick.py --time-only 2>&1 | grep -v amdgpu
} > gpurun_out/r05_call7.log 2>&1
cat gpurun_out/r05_call7.log | tail -30
