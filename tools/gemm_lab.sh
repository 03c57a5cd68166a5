#!/bin/bash
cd $GRAFT_REPO_ROOT
for L in hip nodma nobar nodmabar; do
  echo "== libtaco_$L"
  TACO_LIB=$PWD/tacotron_amd/libtaco_$L.so python tools/gemm_variants.py 32x2 32x3 2>&1 | grep -v amdgpu.ids
done
