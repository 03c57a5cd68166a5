#!/bin/bash
mkdir -p gpurun_out
{
echo "== knob sweep on the wave-uniform poll build"; timeout 600 bash tools/ab_run.sh hip sleep es0 es1 es4 nobs noshadow mvb4
} > gpurun_out/r05_call37.log 2>&1
cat gpurun_out/r05_call37.log | tail -70
