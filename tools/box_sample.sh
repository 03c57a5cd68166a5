#!/bin/bash
# samples the box kind: short step timing; on a fast-kind box (step < 8.95 ms) also the full bench line
cd $GRAFT_REPO_ROOT
python tools/dec_quick.py --time-only 2>&1 | grep -v amdgpu.ids | head -3 > gpurun_out/box_sample.txt
cat gpurun_out/box_sample.txt
S=$(grep -o "S1: [0-9.]*" gpurun_out/box_sample.txt | cut -d' ' -f2)
if python -c "import sys; sys.exit(0 if float('$S') < 8.95 else 1)"; then
  python bench.py > gpurun_out/r04_bench_fastbox.json 2> gpurun_out/r04_bench_fastbox.err
  python tools/family_trace.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_family_trace_fastbox.txt
  tail -c 300 gpurun_out/r04_bench_fastbox.json
fi
