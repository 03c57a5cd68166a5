export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
python -c "from tacotron_amd import lib; print('clock probe GHz:', [round(lib.clock_probe(), 3) for _ in range(3)])" > $O/r04_clock.txt 2>&1
bash tools/profile_round.sh r04 2>&1 | tail -12
cat $O/r04_clock.txt
