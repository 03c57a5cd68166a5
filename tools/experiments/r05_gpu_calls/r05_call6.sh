#!/bin/bash
mkdir -p gpurun_out
{
echo "== feeder probe"; timeout 300 python tools/feeder_probe.py 2>&1 | grep -v amdgpu.ids
echo "== host loop"; (cd /tmp && timeout 300 python -m tacotron_amd.train --steps 210 -d 1 2>&1 | grep -v amdgpu.ids | tail -3)
echo "== gpu tests"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
echo "== fabric"; python -c "
from tacotron_amd import lib
print(lib.fabric_probe())"
} > gpurun_out/r05_call6.log 2>&1
cat gpurun_out/r05_call6.log | tail -40
