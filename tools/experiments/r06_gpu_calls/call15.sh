#!/bin/bash
# round 6, GPU call 15: two-accumulator bf16x3 as the default of both kernels: op parity, then family traces product vs (TN on one accumulator) vs (NN on one accumulator)
mkdir -p gpurun_out
{
echo "== ops"; timeout 900 python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | grep -v amdgpu.ids | tail -3
echo "== adversarial (printed)"; timeout 600 python -m pytest tests/test_gpu_ops.py -q -s -k "adversarial or chain_bound" 2>&1 | grep -E "rel-L2|passed|failed"
for rep in 1 2 3; do
for l in hip tnacc1 nnacc1; do echo -n "$l: "; TACO_LIB=$PWD/tacotron_amd/libtaco_$l.so timeout 300 python tools/family_trace.py 2>&1 | grep -E "^step|^sum" | tr '\n' ' '; echo; done
done
echo "== tn lines (product vs tnacc1)"
timeout 300 python tools/family_trace.py 2>&1 | grep -E " tn" | cut -c 1-120
TACO_LIB=$PWD/tacotron_amd/libtaco_tnacc1.so timeout 300 python tools/family_trace.py 2>&1 | grep -E " tn" | cut -c 1-120
} > gpurun_out/r06_call15.log 2>&1
cat gpurun_out/r06_call15.log
