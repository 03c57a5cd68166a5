export TMPDIR=/tmp; R=$PWD
python bench.py > gpurun_out/bench_v2.json 2> gpurun_out/bench_v2.err
cd /tmp
timeout 280 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o r -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /tmp/b1.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -d /tmp/p2 -o r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-inference > /tmp/b2.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -d /tmp/p3 -o r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-inference > /tmp/b3.log 2>&1
cd $R
python tools/rocpd_summary.py $(find /tmp/p1 -name "*.db" | head -1) > gpurun_out/kernel_stats_v2.txt 2>&1
python tools/pmc_to_json.py $(find /tmp/p2 -name "*.db" | head -1) $(find /tmp/p3 -name "*.db" | head -1) gpurun_out/pmc_v2.json "round-1 v2: folded decoder rounds (9 fwd / 10 bwd), cross-round weight prefetch" > /dev/null 2> gpurun_out/pmc.err
tail -c 600 gpurun_out/bench_v2.json
