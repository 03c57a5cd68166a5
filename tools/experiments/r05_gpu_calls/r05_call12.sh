#!/bin/bash
mkdir -p gpurun_out
{
for b in 3072 2048 2560 4096 1536 3072; do echo "== TACO_TN_BLOCKS=$b"; TACO_TN_BLOCKS=$b timeout 300 python tools/family_trace.py 2>&1 | grep -E "^step|^sum| tn" | awk '{ if ($1=="step"||$1=="sum") print; else print $1, $2, $3 }' | tr '\n' ' '; echo; done
} > gpurun_out/r05_call12.log 2>&1
cat gpurun_out/r05_call12.log
