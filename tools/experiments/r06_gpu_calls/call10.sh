#!/bin/bash
# round 6, GPU call 10: does the image form move the break-even tile count of the DMA kernel? (TACO_GEMM2_MIN_TILES sweep, family traces)
mkdir -p gpurun_out
{
for mt in 160 88 48 160 88 48; do
  echo "== TACO_GEMM2_MIN_TILES=$mt"; TACO_GEMM2_MIN_TILES=$mt timeout 300 python tools/family_trace.py 2>&1 | grep -E "^step|^sum|K=768|N=256 K=512|N=80 K=256|N=256 K=80 taps=3|N=160|N=256 K=256 taps" | cut -c 1-110
done
} > gpurun_out/r06_call10.log 2>&1
cat gpurun_out/r06_call10.log
