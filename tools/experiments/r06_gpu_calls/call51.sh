#!/bin/bash
# round 6, GPU call 51: forward with the embedding launch in front of the side stream's fork vs behind it (TACO_FWD_FORK_FIRST=1), same box
mkdir -p gpurun_out
{
for i in 1 2 3 4; do
for v in "TACO_NOP=1" "TACO_FWD_FORK_FIRST=1"; do
env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('$v', round(d['ms_per_step'],3), round(k.get('us_per_decoder_step_fwd'),2), round(k.get('us_per_decoder_step_bwd'),2), 'non-decoder', round(d['ms_per_step']-0.18*(k.get('us_per_decoder_step_fwd')+k.get('us_per_decoder_step_bwd')),3), d['box']['shader_clock_ghz_latency_bound'])"
done; done
} > gpurun_out/r06_call51.log 2>&1
cat gpurun_out/r06_call51.log
