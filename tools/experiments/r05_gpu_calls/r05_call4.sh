#!/bin/bash
# round 5, GPU call 4: clocks / power under the bf16x3 kernel vs the fp32 form; one-step timeline of the current build
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
{
echo "== power probe, bf16x3 (default)"; timeout 200 python tools/power_probe.py 2>&1 | grep -v amdgpu.ids
echo "== power probe, fp32 MFMA form"; TACO_GEMM2_BF16X=0 timeout 200 python tools/power_probe.py 2>&1 | grep -v amdgpu.ids | head -12
} > gpurun_out/r05_power.txt 2>&1
cd /tmp
timeout 280 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o r -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-extras > /tmp/b1.log 2>&1
cd $R
python tools/rocpd_timeline.py $(find /tmp/p1 -name "*.db" | head -1) 12 > gpurun_out/r05_step_timeline_mid.txt 2>&1
tail -3 /tmp/b1.log | cut -c 1-300
cat gpurun_out/r05_power.txt
