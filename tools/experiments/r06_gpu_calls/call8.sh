#!/bin/bash
# round 6, GPU call 8: ablation lab of the B-image form (BX = 2): what bounds the k-loop now that the B split is gone?
mkdir -p gpurun_out
{
for l in hip lab6_nodma lab6_nomfma lab6_nosplit lab6_nobar lab6_nobread lab6_nodma_nobread; do
  echo "== lib $l"; TACO_LIB=$PWD/tacotron_amd/libtaco_$l.so timeout 200 python tools/gemm_lab6.py 2>&1 | grep -v amdgpu.ids
done
echo "== lib hip, 4 ring stages"; TACO_GEMM2_BI_NS=4 timeout 200 python tools/gemm_lab6.py 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r06_gemm_lab.txt 2>&1
cat gpurun_out/r06_gemm_lab.txt
