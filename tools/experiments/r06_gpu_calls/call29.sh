#!/bin/bash
# round 6, GPU call 29: the decoder build with all latency-chain changes (epilogue preloads, raw poll units, pinned shadows, G1 / G2 / DQ shadows,
# DQ polled by eight waves) vs the build of the round-6 profile set (libtaco_base.so): parity, three alternations, traces
mkdir -p gpurun_out
{
echo "== parity on the new build"
timeout 1700 python -m pytest tests/test_gpu_model.py -x -q 2>&1 | grep -v amdgpu.ids | tail -3
bash tools/ab_run.sh hip base
python tools/dec3_trace.py 32 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r06_call29.log 2>&1
cat gpurun_out/r06_call29.log | tail -70
