#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== old"; TACO_LIB=$PWD/tacotron_amd/libtaco_gruold.so python tools/gemm_steady.py 2>&1 | grep -v "amdgpu.ids\|vendor"
echo "== new"; python tools/gemm_steady.py 2>&1 | grep -v "amdgpu.ids\|vendor"
