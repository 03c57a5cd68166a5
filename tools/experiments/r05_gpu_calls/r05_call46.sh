#!/bin/bash
mkdir -p gpurun_out
{ echo "== publish without the scalar branch on X.fast (timing probe; every cluster is XCD-local on this box)"; timeout 300 bash tools/ab_run.sh hip alwaysfast; } > gpurun_out/r05_call46.log 2>&1
cat gpurun_out/r05_call46.log
