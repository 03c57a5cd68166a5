#!/bin/bash
# round 6, GPU call 48: k-split for the bi-GRU x-projection input gradient (and, as an experiment, d values): parity subset, A/B
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_gpu_model.py -x -q -k "golden or full_size or trajectory or speaker or optional_paths" 2>&1 | grep -v amdgpu.ids | tail -3
for i in 1 2 3; do
for v in "TACO_XPROJ_BWD_KSPLIT=1" "TACO_XPROJ_BWD_KSPLIT=0" "TACO_DVAL_KSPLIT=1"; do
env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('$v', round(d['ms_per_step'],3), round(k.get('us_per_decoder_step_fwd'),2), round(k.get('us_per_decoder_step_bwd'),2), 'non-decoder', round(d['ms_per_step']-0.18*(k.get('us_per_decoder_step_fwd')+k.get('us_per_decoder_step_bwd')),3))"
done; done
python tools/family_trace.py 2>&1 | grep -v amdgpu.ids | grep -n "K=768\|ksplit\|K=512\|^sum\|^step"
} > gpurun_out/r06_call48.log 2>&1
cat gpurun_out/r06_call48.log
