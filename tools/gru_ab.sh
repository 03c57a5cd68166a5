#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2; do
TACO_LIB=$PWD/tacotron_amd/libtaco_gruold.so python tools/gru_quick.py 2>&1 | tail -1
python tools/gru_quick.py 2>&1 | tail -1
done
python -m pytest tests/test_gpu_ops.py -q -m gpu -k "bigru" 2>&1 | tail -3
python -m pytest tests/test_gpu_model.py -q -m gpu -x 2>&1 | tail -3
