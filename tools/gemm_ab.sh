#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py -q -m gpu -x 2>&1 | tail -3
python tools/gemm_variants.py 32x2 2>&1 | grep -v amdgpu.ids
for i in 1 2; do
TACO_LIB=$PWD/tacotron_amd/libtaco_gruold.so python tools/gru_quick.py 2>&1 | tail -1
python tools/gru_quick.py 2>&1 | tail -1
done
python -m pytest tests/test_gpu_model.py tests/test_gpu_sizes.py -q -m gpu -x 2>&1 | tail -3
