python -m pytest tests -q -m gpu -n 1 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/profile_round.sh > /dev/null 2>&1
head -c 400 gpurun_out/bench_v2.json
