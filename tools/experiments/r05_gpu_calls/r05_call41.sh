#!/bin/bash
# decoder.hip (the fallback) with wave-uniform poll loops: parity of everything that runs it, then timing old vs new at S1 (TACO_DEC_V3=0)
mkdir -p gpurun_out
{
echo "== tests that run decoder.hip (cluster widths, fall-back, escalation, sizes beyond decoder3's scope)"
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_sizes.py -x -q -k "cluster or fall or escal or optional or sizes or beyond or wide or long or b64 or decoder" 2>&1 | tail -4
echo "== timing, TACO_DEC_V3=0"
for rep in 1 2 3; do for n in hip olddec; do echo -n "$n: "; TACO_DEC_V3=0 TACO_QUIET=1 TACO_LIB=$PWD/tacotron_amd/libtaco_$n.so python tools/dec_quick.py --time-only 2>&1 | grep "^S1"; done; done
} > gpurun_out/r05_call41.log 2>&1
cat gpurun_out/r05_call41.log | tail -20
