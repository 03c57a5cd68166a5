export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
python tools/dec_quick.py > $O/r04d_dec_quick.txt 2>&1
bash tools/ab_run.sh hip nors nop128 base > $O/r04d_ab.txt 2>&1
(timeout 900 python -m pytest tests/test_gpu_model.py "tests/test_gpu_sizes.py::test_config1_peaked_attention_full_size" "tests/test_gpu_sizes.py::test_config0_arctic_plumbing_shape" "tests/test_gpu_sizes.py::test_config3_free_running_inference_full_length" -m gpu -x -q 2>&1 | tail -30) > $O/r04d_pytest.txt
cat $O/r04d_ab.txt; tail -3 $O/r04d_pytest.txt; head -20 $O/r04d_dec_quick.txt
