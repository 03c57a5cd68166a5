#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
cd /tmp
timeout 200 rocprofv3 --pmc FETCH_SIZE -d /tmp/pf -o r -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-inference --no-extras > /tmp/bf.log 2>&1
python $R/tools/pmc_per_dispatch.py $(find /tmp/pf -name "*.db" | head -1) FETCH_SIZE conv_gemm2 > $R/gpurun_out/r05_call44.log 2>&1
tail -5 /tmp/bf.log >> $R/gpurun_out/r05_call44.log
tail -40 $R/gpurun_out/r05_call44.log
