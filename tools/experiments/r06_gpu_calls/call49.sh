#!/bin/bash
# round 6, GPU call 49: tail events -- forks back to the joined stream keep the tail; profiling brackets ride on the bracketed launch
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_gpu_model.py -x -q -k "golden or full_size or trajectory or speaker or capturable or optional_paths or tail_event or cluster_geometries" 2>&1 | grep -v amdgpu.ids | tail -3
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_layout.py -x -q 2>&1 | grep -v amdgpu.ids | tail -3
for i in 1 2 3; do
for v in 1 0; do
TACO_TAIL_EVENTS=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('TACO_TAIL_EVENTS=$v', round(d['ms_per_step'],3), round(k.get('us_per_decoder_step_fwd'),2), round(k.get('us_per_decoder_step_bwd'),2), 'non-decoder', round(d['ms_per_step']-0.18*(k.get('us_per_decoder_step_fwd')+k.get('us_per_decoder_step_bwd')),3))"
done; done
export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout 280 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o r -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-extras > /tmp/b1.log 2>&1
cd $R
python tools/rocpd_timeline.py $(find /tmp/p1 -name "*.db" | head -1) 12 > gpurun_out/r06d_step_timeline.txt 2>&1
python tools/rocpd_summary.py $(find /tmp/p1 -name "*.db" | head -1) 2>&1 | grep "decoder3" 
} > gpurun_out/r06_call49.log 2>&1
cat gpurun_out/r06_call49.log
