#!/bin/bash
# round 6, GPU call 45: five gradient segments (the encoder conv bank announced behind its weight-gradient launch, ahead of the step's tail)
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_layout.py -x -q 2>&1 | grep -v amdgpu.ids | tail -3
timeout 1200 python -m pytest tests/test_gpu_model.py -x -q -k "golden or full_size or trajectory or speaker" 2>&1 | grep -v amdgpu.ids | tail -2
TACO_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline --no-inference --no-extras > gpurun_out/r06_bench_forced_dist.json 2> gpurun_out/r06_bench_forced_dist.err
python -c "
import json; d=json.load(open('gpurun_out/r06_bench_forced_dist.json')); print(d['ms_per_step'], json.dumps(d['allreduce']['segments']))"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('plain', d['ms_per_step'])"
python tools/dp_coresidency.py 2>&1 | grep -v amdgpu.ids | tail -4
} > gpurun_out/r06_call45.log 2>&1
cat gpurun_out/r06_call45.log
