#!/bin/bash
mkdir -p gpurun_out
{
for args in "--no-cpu-baseline" "--no-cpu-baseline --no-inference"; do
  echo "-- bench $args"; python bench.py $args 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('S1', j['ms_per_step'], 'vctk', j['vctk']['ms_per_step'], 's2', j['s2']['ms_per_step'])"
done
} > gpurun_out/r06_call7.log 2>&1
cat gpurun_out/r06_call7.log
