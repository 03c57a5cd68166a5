#!/bin/bash
# round 6, GPU call 28: BPTT round DQ's gather polled by all eight waves (loader waves wait for their prefetch first) vs by waves 0-3
mkdir -p gpurun_out
{
bash tools/ab_run.sh hip allpoll base
TACO_LIB=$PWD/tacotron_amd/libtaco_allpollp.so python tools/dec3_trace.py 32 2>&1 | grep -v amdgpu.ids | sed -n '/BACKWARD/,$p'
} > gpurun_out/r06_call28.log 2>&1
cat gpurun_out/r06_call28.log
