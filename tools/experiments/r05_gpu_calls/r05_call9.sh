#!/bin/bash
mkdir -p gpurun_out
{
echo "== traj + feeder + probe tests"; timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_frontend.py -q -s -k "trajectory or feeder or fabric" 2>&1 | grep -E "20 steps|fabric probe|passed|failed|Error|assert" | head -20
for mt in 160 96 64 40; do echo "== family trace MIN_TILES=$mt"; TACO_GEMM2_MIN_TILES=$mt timeout 300 python tools/family_trace.py 2>&1 | grep -E "^step|^sum|^# ?[0-9]+ .* nn " | awk '{ if ($1=="step"||$1=="sum") print; else if ($3+0 < 70 || 1) print }' | grep -E "^step|^sum|N=80 |K=80 |N=128 K=128 taps=3|N=256 K=256 taps=1|N=128 K=768|N=256 K=512|N=160|N=128 K=256 taps|N=256 K=128 taps=1..1$"; done
} > gpurun_out/r05_call9.log 2>&1
cat gpurun_out/r05_call9.log | tail -90
