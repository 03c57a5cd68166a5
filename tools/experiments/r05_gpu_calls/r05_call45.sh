#!/bin/bash
# conv-bank order (problems dealt to XCDs): parity, time (family trace, alternating), fetch bytes per conv_gemm2 launch
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
{
echo "== ops tests + encoder / post-net stage tests"; timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q -k "ops or bank or pool or stage or encoder or cbhg or medium" 2>&1 | tail -3
for x in 1 0 1 0 1 0; do echo "== family trace TACO_GEMM2_BANK_XCD=$x"; TACO_GEMM2_BANK_XCD=$x timeout 300 python tools/family_trace.py 2>&1 | grep -E "^step|^sum|pool1"; done
cd /tmp
for x in 1 0; do
TACO_GEMM2_BANK_XCD=$x timeout 200 rocprofv3 --pmc FETCH_SIZE -d /tmp/pf$x -o r -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-inference --no-extras > /tmp/bf.log 2>&1
echo "== FETCH_SIZE per conv_gemm2 launch, TACO_GEMM2_BANK_XCD=$x"; python $R/tools/pmc_per_dispatch.py $(find /tmp/pf$x -name "*.db" | head -1) FETCH_SIZE conv_gemm2 | head -14
done
} > gpurun_out/r05_call45.log 2>&1
cat gpurun_out/r05_call45.log
