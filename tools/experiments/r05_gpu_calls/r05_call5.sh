#!/bin/bash
# round 5, GPU call 5: whole GPU suite on the current build, bench line with box / ab objects, host loop with the feeder
mkdir -p gpurun_out
{
echo "== gpu tests"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
echo "== host loop (python -m tacotron_amd.train, synthetic corpus, 210 steps)"; (cd /tmp && timeout 300 python -m tacotron_amd.train --steps 210 -d 1 2>&1 | grep -v amdgpu.ids | tail -4)
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/r05_bench_mid.err | tail -1 > gpurun_out/r05_bench_mid.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_bench_mid.json'))
print('ms/step', d['ms_per_step'], 'value', d['value'])
print('box', json.dumps(d['box']))
print('ab', json.dumps(d.get('ab')))
print('kernels', d['kernels_ms'])
print('gemm family', {k: d['rooflines'][1].get(k) for k in ('achieved','frac','ms_per_step_summed','nn_kernel_4096_cubed_tflops','vendor_blas_4096_cubed_tflops')})
print('s2', d['s2']['ms_per_step'], 'vctk', d['vctk']['ms_per_step'], 'inf', d['inference']['B1']['ms_per_batch'], d['inference']['B32']['ms_per_batch'])
PY
tail -3 gpurun_out/r05_bench_mid.err
} > gpurun_out/r05_call5.log 2>&1
cat gpurun_out/r05_call5.log | tail -40
