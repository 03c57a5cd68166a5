# Round profile on the GPU box: bench line, rocprofv3 kernel stats + one-step timeline, PMC passes (FETCH / WRITE / MFMA), probes.
# usage: bash tools/profile_round.sh r02   -> files under gpurun_out/<tag>_*
TAG=${1:-r04}; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
H=$(python -c "import bench; print(bench.source_hash())")
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
cd /tmp
SHORT="--no-cpu-baseline --no-inference --no-extras"
timeout 280 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o r -- python $R/bench.py --steps 20 --warmup 5 $SHORT > /tmp/b1.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -d /tmp/p2 -o r -- python $R/bench.py --steps 2 --warmup 1 $SHORT > /tmp/b2.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -d /tmp/p3 -o r -- python $R/bench.py --steps 2 --warmup 1 $SHORT > /tmp/b3.log 2>&1
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -d /tmp/p4 -o r -- python $R/bench.py --steps 2 --warmup 1 $SHORT > /tmp/b4.log 2>&1
cd $R
python tools/rocpd_summary.py $(find /tmp/p1 -name "*.db" | head -1) > $O/${TAG}_kernel_stats.txt 2>&1
python tools/rocpd_timeline.py $(find /tmp/p1 -name "*.db" | head -1) 12 > $O/${TAG}_step_timeline.txt 2>&1
python tools/pmc_to_json.py $(find /tmp/p2 -name "*.db" | head -1) $(find /tmp/p3 -name "*.db" | head -1) $O/${TAG}_pmc.json $H "round $TAG" > /dev/null 2> $O/${TAG}_pmc.err
python tools/pmc_mfma.py $(find /tmp/p4 -name "*.db" | head -1) $H > $O/${TAG}_pmc_mfma.txt 2>&1
python tools/pmc_step.py $(find /tmp/p2 -name "*.db" | head -1) $(find /tmp/p3 -name "*.db" | head -1) 3 $O/${TAG}_pmc_step.json $H > $O/${TAG}_pmc_step.txt 2>&1
python - <<PY
import json
a = json.load(open('$O/${TAG}_pmc.json')); a['step'] = json.load(open('$O/${TAG}_pmc_step.json'))
json.dump(a, open('$O/${TAG}_pmc.json', 'w'), indent=1)
PY
python tools/dec3_trace.py 32 > $O/${TAG}_dec_probes.txt 2>&1
python tools/dec3_trace.py 1 infer >> $O/${TAG}_dec_probes.txt 2>&1
(export TACO_LIB=$R/tacotron_amd/libtaco_probe.so; python tools/dec_quick.py --time-only; TACO_DEC_FAKEX=1 python tools/dec_quick.py --time-only; TACO_DEC_V3_AGENT=1 python tools/dec_quick.py --time-only) >> $O/${TAG}_dec_probes.txt 2>&1
python tools/family_trace.py > $O/${TAG}_family_trace.txt 2>&1
TACO_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline --no-inference --no-extras > $O/${TAG}_bench_forced_dist.json 2> $O/${TAG}_bench_forced_dist.err
python tools/dp_coresidency.py > $O/${TAG}_dp_coresidency.txt 2>&1
python tools/soak.py > $O/${TAG}_soak.txt 2>&1
tail -c 300 $O/${TAG}_bench.json; head -12 $O/${TAG}_pmc_mfma.txt
