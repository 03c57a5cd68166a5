#!/bin/bash
# XCD-aware block order of the weight-gradient (TN) launches: parity, time (family trace), L2 -> fabric fetch bytes (PMC)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
{
echo "== ops tests (tn)"; timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "tn or TN or weight" 2>&1 | tail -3
for x in 1 0 1 0; do echo "== family trace TACO_TN_XCD=$x"; TACO_TN_XCD=$x timeout 300 python tools/family_trace.py 2>&1 | grep -E "^step|^sum| tn"; done
cd /tmp
for x in 1 0; do
  TACO_TN_XCD=$x timeout 200 rocprofv3 --pmc FETCH_SIZE -d /tmp/px$x -o r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-inference --no-extras > /tmp/bx$x.log 2>&1
  echo "== FETCH_SIZE TACO_TN_XCD=$x"; python $R/tools/pmc_generic.py $(find /tmp/px$x -name "*.db" | head -1) gemm_tn
done
} > gpurun_out/r05_call42.log 2>&1
cat gpurun_out/r05_call42.log | tail -120
