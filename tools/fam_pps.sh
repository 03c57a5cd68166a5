#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "shifted" 2>&1 | tail -2
for L in hip pps2 pps4 hip; do
TACO_LIB=$PWD/tacotron_amd/libtaco_$L.so python tools/family_trace.py 2>&1 | grep -v amdgpu.ids > gpurun_out/fam_$L.txt
done
paste <(cut -c1-18 gpurun_out/fam_pps2.txt) <(cut -c1-18 gpurun_out/fam_pps4.txt) <(cut -c1-120 gpurun_out/fam_hip.txt) | grep -E "nn |step|sum"
