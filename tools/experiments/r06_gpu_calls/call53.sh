#!/bin/bash
# round 6, GPU call 53: NN-kernel routing knobs re-measured on the final build (step time, three alternations)
mkdir -p gpurun_out
{
for i in 1 2 3; do
for v in "TACO_NOP=1" "TACO_GEMM2_MIN_TILES=96" "TACO_GEMM2_MIN_TILES=256" "TACO_GEMM2_BI_NS=4" "TACO_GEMM2_BI_NS=2"; do
env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('$v', round(d['ms_per_step'],3), 'non-decoder', round(d['ms_per_step']-0.18*(k.get('us_per_decoder_step_fwd')+k.get('us_per_decoder_step_bwd')),3))"
done; done
} > gpurun_out/r06_call53.log 2>&1
cat gpurun_out/r06_call53.log
