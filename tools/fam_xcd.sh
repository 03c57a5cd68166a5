#!/bin/bash
cd $GRAFT_REPO_ROOT
TACO_GEMM2_XCD=0 python tools/family_trace.py 2>&1 | grep -v amdgpu.ids > gpurun_out/fam_xcd0.txt
python tools/family_trace.py 2>&1 | grep -v amdgpu.ids > gpurun_out/fam_xcd1.txt
paste <(cut -c1-18 gpurun_out/fam_xcd0.txt) <(cut -c1-120 gpurun_out/fam_xcd1.txt) | grep -E "nn |step|sum"
python tools/dense_probe.py 2>&1 | grep -v amdgpu
python -m pytest tests/test_gpu_ops.py -q -m gpu -x 2>&1 | tail -2
