#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py -q -m gpu -x 2>&1 | tail -2
TACO_LIB=$PWD/tacotron_amd/libtaco_gruold.so python tools/family_trace.py 2>&1 | grep -v amdgpu.ids > gpurun_out/fam_old.txt
python tools/family_trace.py 2>&1 | grep -v amdgpu.ids > gpurun_out/fam_new.txt
paste <(cut -c1-18 gpurun_out/fam_old.txt) <(cut -c1-110 gpurun_out/fam_new.txt) | grep -E " tn|step|sum"
python -m pytest tests/test_gpu_model.py -q -m gpu -x 2>&1 | tail -2
