export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
(timeout 1200 python -m pytest "tests/test_gpu_sizes.py::test_config4_vctk_109_speakers_full_size" "tests/test_gpu_sizes.py::test_encoder_bottom_gradient_deviation_is_localized" "tests/test_gpu_sizes.py::test_config1_peaked_attention_full_size" tests/test_gpu_dist.py "tests/test_gpu_model.py::test_error_words_are_sticky_and_guard_the_update" -m gpu -q -s 2>&1 | grep -v "^  grad .*e-0[5-9]$") > $O/r04c_pytest.txt
tail -3 $O/r04c_pytest.txt
