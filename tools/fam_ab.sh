#!/bin/bash
cd $GRAFT_REPO_ROOT
TACO_LIB=$PWD/tacotron_amd/libtaco_gruold.so python tools/family_trace.py 2>&1 | grep -v amdgpu.ids > gpurun_out/fam_old.txt
python tools/family_trace.py 2>&1 | grep -v amdgpu.ids > gpurun_out/fam_new.txt
paste <(cut -c1-28 gpurun_out/fam_old.txt) <(cut -c1-200 gpurun_out/fam_new.txt)
