#!/bin/bash
# round 5, GPU call 3: bf16x3 with the B tile split once per workgroup (BX = 2): parity, lab builds, family trace vs BX = 1
mkdir -p gpurun_out
{
echo "== ops"; timeout 600 python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -5
echo "== NN kernel, plain shapes, steady state: bx2 16x3 | bx2 16x4 (1 WG/CU) | bx1 16x4 | f32"
TACO_GEMM2_BF16X=2 timeout 120 python tools/gemm_variants.py 16x3 16x4 2>&1 | grep -v amdgpu.ids
TACO_GEMM2_BF16X=1 timeout 120 python tools/gemm_variants.py 16x4 2>&1 | grep -v amdgpu.ids
for l in lab_nosplit lab_nomfma lab_nodma lab_nosplitnodma; do echo "-- lib $l (bx2 16x3)"; TACO_LIB=$PWD/tacotron_amd/libtaco_$l.so timeout 120 python tools/gemm_variants.py 16x3 2>&1 | grep -v amdgpu.ids; done
echo "== family trace BX=2"; TACO_GEMM2_BF16X=2 timeout 300 python tools/family_trace.py 2>&1 | grep -v amdgpu.ids
echo "== family trace BX=1"; TACO_GEMM2_BF16X=1 timeout 300 python tools/family_trace.py 2>&1 | grep -E "^step|^sum| nn.*(pool|ksplit|gather|N=1025|K=1028)"
echo "== model tests"; timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_sizes.py -x -q 2>&1 | tail -5
} > gpurun_out/r05_call3.log 2>&1
tail -c 2500 gpurun_out/r05_call3.log
