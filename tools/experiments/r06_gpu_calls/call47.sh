#!/bin/bash
# round 6, GPU call 47: tail events with learned launch plans (events only on the launches a fork / join consumes): parity subset, A/B
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_gpu_model.py -x -q -k "golden or full_size or trajectory or speaker or capturable or optional_paths" 2>&1 | grep -v amdgpu.ids | tail -3
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_layout.py -x -q 2>&1 | grep -v amdgpu.ids | tail -3
for i in 1 2 3; do
for v in 1 0 2; do
TACO_TAIL_EVENTS=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('TACO_TAIL_EVENTS=$v', round(d['ms_per_step'],3), round(d['kernels_ms'].get('us_per_decoder_step_fwd'),2), round(d['kernels_ms'].get('us_per_decoder_step_bwd'),2))"
done; done
} > gpurun_out/r06_call47.log 2>&1
cat gpurun_out/r06_call47.log
