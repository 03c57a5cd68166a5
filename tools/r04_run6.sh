export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
bash tools/ab_run.sh hip oldfandq noburst > $O/r04f_ab.txt 2>&1
for n in hip noburst hip noburst; do echo "== $n"; TACO_LIB=$R/tacotron_amd/libtaco_$n.so python tools/family_trace.py 2>&1 | grep -E "^step|^sum|^bigru"; done > $O/r04f_bigru.txt 2>&1
(timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -5) > $O/r04f_pytest.txt
cat $O/r04f_ab.txt $O/r04f_bigru.txt; tail -3 $O/r04f_pytest.txt
