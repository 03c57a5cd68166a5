export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
H=$(python -c "import bench; print(bench.source_hash())")
python -c "from tacotron_amd import lib; print('clock probe GHz (chip-wide latency-bound load):', [round(lib.clock_probe(), 3) for _ in range(3)])" > $O/r04_clock.txt 2>&1
python tools/dec_quick.py --time-only >> $O/r04_clock.txt 2>&1
cd /tmp
SHORT="--no-cpu-baseline --no-inference --no-extras"
timeout 200 rocprofv3 --pmc FETCH_SIZE -d /tmp/p2 -o r -- python $R/bench.py --steps 2 --warmup 1 $SHORT > /tmp/b2.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -d /tmp/p3 -o r -- python $R/bench.py --steps 2 --warmup 1 $SHORT > /tmp/b3.log 2>&1
cd $R
python tools/pmc_to_json.py $(find /tmp/p2 -name "*.db" | head -1) $(find /tmp/p3 -name "*.db" | head -1) $O/r04_pmc.json $H "round r04" > /dev/null 2> $O/r04_pmc.err
python tools/pmc_step.py $(find /tmp/p2 -name "*.db" | head -1) $(find /tmp/p3 -name "*.db" | head -1) 3 $O/r04_pmc_step.json $H > $O/r04_pmc_step.txt 2>&1
python - <<PY
import json
a = json.load(open('$O/r04_pmc.json')); a['step'] = json.load(open('$O/r04_pmc_step.json'))
json.dump(a, open('$O/r04_pmc.json', 'w'), indent=1)
PY
cp $O/r04_pmc.json $R/profiles/pmc_latest.json
python bench.py > $O/r04_bench.json 2> $O/r04_bench.err
cat $O/r04_clock.txt $O/r04_pmc_step.txt; tail -c 400 $O/r04_bench.json
