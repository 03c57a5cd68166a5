#!/bin/bash
# round 6, GPU call 4: forward weight images built on the side stream (event wait in front of the encoder CBHG): parity + S1 A/B + full bench line
mkdir -p gpurun_out
{
echo "== tests"; timeout 2400 python -m pytest tests/test_gpu_model.py tests/test_gpu_sizes.py tests/test_gpu_ops.py -x -q 2>&1 | grep -v amdgpu.ids | tail -5
for v in 0 1 0 1; do echo -n "BSPLIT=$v: "; TACO_GEMM2_BSPLIT=$v timeout 300 python tools/dec_quick.py --time-only 2>&1 | grep "^S1"; done
echo "== bench"; timeout 900 python bench.py 2>gpurun_out/r06_bench4.err | tail -1 > gpurun_out/r06_bench4.json; python -c "
import json; j=json.load(open('gpurun_out/r06_bench4.json')); print({k: j[k] for k in ('value','ms_per_step','kernels_ms')}); print(j['rooflines'][1]['ms_per_step_summed'], j['ab'], j['vctk']['ms_per_step'], j['s2']['ms_per_step'], j['inference']['B1']['ms_per_batch'], j['inference']['B32']['ms_per_batch'])"
} > gpurun_out/r06_call4.log 2>&1
tail -c 5000 gpurun_out/r06_call4.log
