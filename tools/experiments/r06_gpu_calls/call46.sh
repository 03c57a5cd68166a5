#!/bin/bash
# round 6, GPU call 46: tail events (stop event on the producing launch instead of a recorded marker): parity subset, A/B, timeline
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_gpu_model.py -x -q -k "golden or full_size or trajectory or speaker or capturable or optional_paths" 2>&1 | grep -v amdgpu.ids | tail -3
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_layout.py -x -q 2>&1 | grep -v amdgpu.ids | tail -3
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tail events on ', d['ms_per_step'], d['kernels_ms'].get('us_per_decoder_step_fwd'), d['kernels_ms'].get('us_per_decoder_step_bwd'))"
TACO_TAIL_EVENTS=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tail events off', d['ms_per_step'], d['kernels_ms'].get('us_per_decoder_step_fwd'), d['kernels_ms'].get('us_per_decoder_step_bwd'))"
done
export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout 280 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o r -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-extras > /tmp/b1.log 2>&1
cd $R
python tools/rocpd_timeline.py $(find /tmp/p1 -name "*.db" | head -1) 12 > gpurun_out/r06b_step_timeline.txt 2>&1
tail -5 gpurun_out/r06b_step_timeline.txt
} > gpurun_out/r06_call46.log 2>&1
cat gpurun_out/r06_call46.log
