#!/bin/bash
mkdir -p gpurun_out
{
echo "== gpu tests"; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
} > gpurun_out/r05_call40.log 2>&1
timeout 1500 bash tools/profile_round.sh r05 > gpurun_out/r05_call40_profile.log 2>&1
cat gpurun_out/r05_call40.log | tail -20; tail -5 gpurun_out/r05_call40_profile.log
