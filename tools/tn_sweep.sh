#!/bin/bash
# Sweep of the weight-gradient launch's workgroup target (TACO_TN_BLOCKS) on the GPU box; per-launch family traces.
mkdir -p gpurun_out
for v in ${TN_SWEEP:-384 768 1536 3072 6144}; do
  echo "== TACO_TN_BLOCKS=$v"
  TACO_TN_BLOCKS=$v python tools/family_trace.py > gpurun_out/tn_sweep_$v.txt 2>&1
  grep -E "^step|^sum" gpurun_out/tn_sweep_$v.txt
done
