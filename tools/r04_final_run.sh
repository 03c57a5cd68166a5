export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
python -m pytest tests -q -m gpu -x > $O/r04_parity.txt 2>&1; tail -3 $O/r04_parity.txt
python -c "from tacotron_amd import lib; print('clock probe GHz:', [round(lib.clock_probe(), 3) for _ in range(3)])" > $O/r04_clock.txt 2>&1
bash tools/profile_round.sh r04 2>&1 | tail -12
python tools/gemm_steady.py 2>&1 | grep -v amdgpu.ids > $O/r04_gemm_steady.txt
python tools/gemm_shapes.py 2>&1 | grep -v amdgpu.ids > $O/r04_gemm_shapes_final.txt
python tools/dense_probe.py 2>&1 | grep -v amdgpu.ids > $O/r04_dense_probe.txt
python tools/gru_quick.py 2>&1 | tail -1 > $O/r04_gru_quick.txt
bash tools/infer_tl.sh > /dev/null 2>&1
cat $O/r04_clock.txt
