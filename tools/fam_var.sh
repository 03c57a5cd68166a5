#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-16x4 16x5 32x2}; do
TACO_GEMM2_VARIANT=$v python tools/family_trace.py 2>&1 | grep -v amdgpu.ids > gpurun_out/fam_$v.txt
done
paste <(cut -c1-18 gpurun_out/fam_16x4.txt) <(cut -c1-18 gpurun_out/fam_16x5.txt) <(cut -c1-120 gpurun_out/fam_32x2.txt)
