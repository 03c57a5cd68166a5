#!/bin/bash
# round 6, GPU call 50: the PMC passes again with tools/pmc_step.py's set-up launches kept out of the per-step traffic
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; TAG=r06h
H=$(python -c "import bench; print(bench.source_hash())")
cd /tmp
SHORT="--no-cpu-baseline --no-inference --no-extras"
timeout 200 rocprofv3 --pmc FETCH_SIZE -d /tmp/p2 -o r -- python $R/bench.py --steps 2 --warmup 1 $SHORT > /tmp/b2.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -d /tmp/p3 -o r -- python $R/bench.py --steps 2 --warmup 1 $SHORT > /tmp/b3.log 2>&1
cd $R
python tools/pmc_to_json.py $(find /tmp/p2 -name "*.db" | head -1) $(find /tmp/p3 -name "*.db" | head -1) $O/${TAG}_pmc.json $H "round $TAG" > /dev/null 2> $O/${TAG}_pmc.err
python tools/pmc_step.py $(find /tmp/p2 -name "*.db" | head -1) $(find /tmp/p3 -name "*.db" | head -1) 3 $O/${TAG}_pmc_step.json $H > $O/${TAG}_pmc_step.txt 2>&1
python - <<PY
import json
a = json.load(open('$O/${TAG}_pmc.json')); a['step'] = json.load(open('$O/${TAG}_pmc_step.json'))
json.dump(a, open('$O/${TAG}_pmc.json', 'w'), indent=1)
PY
cat $O/${TAG}_pmc_step.txt
