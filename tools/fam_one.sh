#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/family_trace.py 2>&1 | grep -v amdgpu.ids > gpurun_out/fam_one.txt
grep -E "^step|^sum| tn" gpurun_out/fam_one.txt | cut -c1-120
for i in 1 2 3; do python tools/gru_quick.py 2>&1 | tail -1 | cut -c1-40; done
python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "tn" 2>&1 | tail -2
